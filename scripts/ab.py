"""A/B two builds of libsigsvgd_hip.so on the SAME GPU box, interleaved (cdna guide rule 24).
usage: python scripts/ab.py libA.so libB.so [rounds]"""
import os, subprocess, sys
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, SIGSVGD_LIB_PATH=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "scripts/dbgbench.py", "c4"], env=env, capture_output=True, text=True).stdout
        print(os.path.basename(lib), out.strip().splitlines()[-1] if out.strip() else "??", flush=True)
