#!/bin/bash
# Register / LDS / occupancy summary of the kernels in one .hip file (compile-time remarks; no GPU needed).
# usage: tools/kernel_resources.sh blasted_amd/csrc/kernels_levelw.hip [name-filter]
F=$1; FILTER=${2:-.}
cd "$(dirname "$F")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -c "$(basename "$F")" -o /tmp/kr_$$.o 2>&1 \
 | awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
        /    VGPRs:/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)}
        /AGPRs:/ {ag=$0; sub(/.*AGPRs: /,"",ag); sub(/ \[.*/,"",ag)}
        /VGPRs Spill:/ {sp=$0; sub(/.*Spill: /,"",sp); sub(/ \[.*/,"",sp)}
        /Occupancy/ {oc=$0; sub(/.*: /,"",oc); sub(/ \[.*/,"",oc)}
        /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-4s vgpr %-4s agpr %-3s spill %-4s occ %-2s lds  %s\n", "", v, ag, sp, oc, l " " name}' \
 | while read -r line; do n=$(echo "$line" | awk '{print $NF}'); echo "${line% *} $(echo $n | c++filt | cut -c1-110)"; done | grep -E "$FILTER"
rm -f /tmp/kr_$$.o
