"""precision_sweep.py restricted to shapes given on the command line: N,T,d,order ..."""
import sys
sys.path.insert(0, "scripts")
import precision_sweep as P
P.SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
P.main(sys.argv[1])
