#!/bin/bash
sample() { for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done; }
echo "== d2d copy loop (4 GiB tensors)"
(python3 - <<'PY'
import torch, time
a = torch.empty(1 << 29, dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
torch.cuda.synchronize(); t = time.time()
n = 0
while time.time() - t < 9:
    for _ in range(50): b.copy_(a)
    torch.cuda.synchronize(); n += 50
print("copy GB/s", n * 2 * a.numel() * 8 / (time.time() - t) / 1e9)
PY
) > gpurun_out/pw_copy.txt 2>&1 &
P=$!; sleep 5; sample; wait $P; cat gpurun_out/pw_copy.txt | tail -1
echo "== butterfly microbenchmark back to back"
(for i in $(seq 1 40); do ./tools/bfly_lab; done > /dev/null 2>&1) &
P=$!; sleep 3; sample; kill $P 2>/dev/null; wait $P 2>/dev/null
