# per-kernel VGPR / scratch / occupancy of one translation unit:  bash tools/kernel_regs.sh vts_wgrad_run.hip [filter]
cd "$(dirname "$0")/../visual-tactile-synthesis_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/kr.o 2>&1 | \
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -e 's/.*remark: [^ ]* *//;s/ *\[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - | \
  sed -e 's/Function Name: //;s/ScratchSize \[bytes\/lane\]/scratch/;s/Occupancy \[waves\/SIMD\]/occ/' | sort -u | (if [ -n "$2" ]; then grep "$2"; else cat; fi) | (if command -v c++filt > /dev/null; then c++filt | sed -e 's/(anonymous namespace):://g;s/void //'; else cat; fi)
