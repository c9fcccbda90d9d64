"""A/B two builds of libsigsvgd_hip.so on the SAME GPU box, interleaved (cdna guide rule 24).
usage: python scripts/ab.py libA.so libB.so [rounds] [c4|c5|stream]"""
import os, subprocess, sys
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
shape = sys.argv[4] if len(sys.argv) > 4 else "c4"
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, SIGSVGD_LIB_PATH=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "scripts/dev/bench_shapes.py", shape], env=env, capture_output=True, text=True).stdout
        print(os.path.basename(lib), " || ".join(out.strip().splitlines()) if out.strip() else "??", flush=True)
