set -o pipefail
mkdir -p gpurun_out/r3t
for n in 100 300 500 1000 1300; do for pfx in 30 166; do for keep in 0 0.5 0.1; do
  v=$((2400000000 / (n * 4 + pfx + 300)))
  extra=""; [ "$keep" != "0" ] && extra="--keep-frac $keep"
  echo "## N=$n prefix=$pfx keep=$keep"; python tools/ab_probe.py --samples $n --variants $v --lines $pfx $extra --arms auto runs pick --rounds 3 2>&1 | grep -v amdgpu.ids | tail -4
done; done; done > gpurun_out/r3t/lines_sweep.log 2>&1
cat gpurun_out/r3t/lines_sweep.log
