"""A/B two builds of libsigsvgd_hip.so on the SAME GPU box, interleaved (cdna guide rule 24).
usage: python scripts/ab.py libA.so libB.so [libC.so ...] [rounds] [c4|c5|stream|short]"""
import os, subprocess, sys
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
rounds = int(rest[0]) if rest else 3
shape = rest[1] if len(rest) > 1 else "c4"
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, SIGSVGD_LIB_PATH=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "scripts/dev/bench_shapes.py", shape], env=env, capture_output=True, text=True).stdout
        print(os.path.basename(lib), " || ".join(out.strip().splitlines()) if out.strip() else "??", flush=True)
