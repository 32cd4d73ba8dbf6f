#!/bin/bash
mkdir -p gpurun_out/repro
TL=$(python -c "import torch,os; print(os.path.dirname(torch.__file__)+'/lib')")
for cfg in "0 0 0" "0 0 1" "1 0 1"; do set -- $cfg; echo "== victim $1 trigger $2 graphs $3 (torch's runtime, preloaded)"; LD_PRELOAD=$TL/libamdhip64.so timeout 300 tools/_race_repro 24 307200 $1 $2 $3 > gpurun_out/repro/torchrt_v$1_t$2_g$3.log 2>&1; echo rc=$?; grep -E "SUMMARY|HIP runtime" gpurun_out/repro/torchrt_v$1_t$2_g$3.log; grep "^   run" gpurun_out/repro/torchrt_v$1_t$2_g$3.log | head -8; done
