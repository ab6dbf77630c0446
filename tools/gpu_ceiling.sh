#!/usr/bin/env bash
set -u
OUT=$PWD/gpurun_out/${1:-ceil}; mkdir -p "$OUT"
timeout 120 tools/_build/mfma_ceiling | tee "$OUT/mfma_ceiling.log"
false && timeout 300 python - <<'PY' | tee "$OUT/matmul.log"
import torch, json
for n in (4096, 8192):
    a = torch.randn(n, n, device="cuda", dtype=torch.float16); b = torch.randn(n, n, device="cuda", dtype=torch.float16)
    for _ in range(3): a @ b
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(json.dumps({"test": "torch_matmul_fp16", "n": n, "ms": round(ms, 4), "tflops": round(2 * n**3 / ms / 1e9, 1)}))
PY
