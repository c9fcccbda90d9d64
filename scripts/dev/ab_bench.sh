#!/bin/bash
# same-box A/B of two builds of the library on the bench's configurations: sigsvgd_amd/_exp/lib_prev.so against the in-tree build
# usage (GPU box): bash scripts/dev/ab_bench.sh
for rep in 1 2; do
  for lib in sigsvgd_amd/_exp/lib_prev.so sigsvgd_amd/libsigsvgd_hip.so; do
    SIGSVGD_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
oc=d.get('other_configs',{})
print('$lib', 'C4 %.4f ms' % d['ms_per_step'], ' '.join('%s=%.4f' % (k.split()[0]+k.split()[1] if k.startswith('ref') else k.split(',')[0], v.get('ms_per_iter', v.get('ms_per_gram_and_gradient',0))) for k,v in oc.items()))"
  done
done
