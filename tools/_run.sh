set -o pipefail
bash tools/profile_bench.sh r03_c3 > gpurun_out/r03_c3.txt 2>&1 || { tail -5 gpurun_out/r03_c3.txt; exit 1; }
tail -4 gpurun_out/r03_c3.txt | cut -c1-300
bash tools/profile_bench.sh r03_c5shard --config c5shard > gpurun_out/r03_c5shard.txt 2>&1 || { tail -5 gpurun_out/r03_c5shard.txt; exit 1; }
tail -5 gpurun_out/r03_c5shard.txt | cut -c1-300
