"""sigsvgd_amd -- MI355X-native signature-kernel SVGD hot path (drop-in for lubaroli/sigsvgd's
`SignatureKernel` / `sigkernel.SigKernel.compute_Gram` / `SVGD.step` surface)."""
__version__ = "0.1.0"
