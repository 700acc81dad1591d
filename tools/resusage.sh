#!/bin/bash
# prints VGPR / SGPR / scratch / occupancy per kernel of a HIP source (tooling)
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/resusage.o 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|TotalSGPRs" \
 | sed -E 's/^.*remark: +//; s/ \[-Rpass.*$//' \
 | awk '/Function Name/{if(line)print line; line=$3} /VGPRs:/{line=line" vgpr="$2} /TotalSGPRs/{line=line" sgpr="$2} /ScratchSize/{line=line" scratch="$3} /Occupancy/{line=line" occ="$3} /LDS Size/{line=line" lds="$4} END{print line}' \
 | while read l; do n=$(echo $l | cut -d' ' -f1 | c++filt | cut -c1-90); echo "$n | $(echo $l | cut -d' ' -f2-)"; done
