#!/bin/bash
# kernel resource usage of one .hip file, one line per kernel:  tools/kres.sh csrc/file.hip [filter]
f=$1; pat=${2:-.}
timeout -k 5 600 /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>/tmp/kres.log
grep -q "error:" /tmp/kres.log && { grep -A3 "error:" /tmp/kres.log | head -40; exit 1; }
awk '/Function Name:/ {name=$NF} /remark:.* VGPRs:/ {v=$NF} /SGPRs:/ && !/Spill/ {sg=$NF} /ScratchSize/ {sc=$NF} /VGPRs Spill/ {sp=$NF} /LDS Size/ {print name, "vgpr", v, "sgpr", sg, "scratch", sc, "spill", sp}' <(sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' /tmp/kres.log) | c++filt | grep -E "$pat"
