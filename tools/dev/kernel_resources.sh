#!/bin/bash
# VGPRs / scratch / LDS of every kernel matching <pattern> in one source file:  kernel_resources.sh <file.hip> <pattern> [flags...]
src=$1; pat=$2; shift 2
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -c /root/repo/pcfa_amd/csrc/$src -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|ScratchSize|LDS Size" \
  | awk '/error/ {print} /Function Name/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)} /VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /LDS Size/ {print v, "vgpr", sc, "scratch", $(NF-1), "lds", name}' \
  | grep -E "$pat" | cut -c1-150
