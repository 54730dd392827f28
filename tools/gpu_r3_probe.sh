#!/bin/bash
# round 3: the issue probe, then the same mix for a few seconds with rocm-smi sampling clock and package power (4 and 3 waves per SIMD)
mkdir -p gpurun_out
tools/issue_probe > gpurun_out/r3_issue_probe.txt 2>&1 || exit 1
for w in 4 3; do
  rm -f gpurun_out/.long gpurun_out/.smi
  (tools/issue_probe long $w > gpurun_out/.long) &
  BP=$!
  sleep 0.8
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | sed 's/.*: //' | tr '\n' ' ' >> gpurun_out/.smi; echo >> gpurun_out/.smi; sleep 0.3; done
  wait $BP
  cat gpurun_out/.long >> gpurun_out/r3_issue_probe.txt
  echo "# sclk / package power while it ran (rocm-smi, 0.3 s apart):" >> gpurun_out/r3_issue_probe.txt
  cat gpurun_out/.smi >> gpurun_out/r3_issue_probe.txt
done
rm -f gpurun_out/.long gpurun_out/.smi
cat gpurun_out/r3_issue_probe.txt
