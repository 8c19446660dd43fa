#!/bin/bash
# Register / scratch / LDS use of every kernel, one line each (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh [kernel-file-stem ...]      (default: all)
cd "$(dirname "$0")/../mp3-enc-bsd_amd/csrc" || exit 1
K=${@:-k_fft k_psy k_fbmdct k_prep k_loop k_format}
for k in $K; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I. -I../../include -Rpass-analysis=kernel-resource-usage -c $k.hip -o /tmp/kr_$k.o 2>&1 |
  awk '/Function Name/ {name=$NF=="[-Rpass-analysis=kernel-resource-usage]"?$(NF-1):$NF}
       /remark: +(VGPRs|AGPRs|TotalSGPRs|ScratchSize|Occupancy|LDS Size)/ {sub(/.*remark: +/,""); sub(/ \[-Rpass.*/,""); line=line "  " $0}
       /LDS Size/ {d=name; sub(/^_Z[0-9]+/,"",d); d=substr(d,1,40); printf "%-42s%s\n", d, line; line=""}'
done
