#!/bin/bash
# compile one csrc file and summarise kernel resource usage: name, VGPRs, spills, occupancy
# usage: scripts/dev/resusage.sh gram_quad.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c sigsvgd_amd/csrc/$f -o /tmp/${f%.hip}.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
  awk '/error|warning:/ {print} /Function Name/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[-R.*/,"",n)} / VGPRs:/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)} /VGPRs Spill/ {sp=$0; sub(/.*Spill: /,"",sp); sub(/ \[.*/,"",sp)} /ScratchSize/ {sc=$0; sub(/.*: /,"",sc); sub(/ \[.*/,"",sc)} /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)} /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-90s vgpr %s spill %s scratch %s occ %s lds %s\n", n, v, sp, sc, o, l}' | c++filt | sed 's/sigsvgd:://g'
