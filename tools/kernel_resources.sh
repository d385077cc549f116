# VGPRs / scratch / occupancy of every kernel variant (cross-compiles; no GPU needed)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$(dirname $0)/../include -Wno-pass-failed --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -c $(dirname $0)/../vgen_amd/csrc/device/kernels.hip -o /tmp/isa/k2.co 2>&1 |
  grep -E "Function Name|VGPRs:|Occupancy|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - |
  sed 's/Function Name: _ZN2vg//; s/ \[bytes\/lane\]//; s/ \[waves\/SIMD\]//'
