#!/bin/bash
# End-to-end config 2 on the GPU box: synthesise the chr22-shape triple on tmpfs, run the CLI, report timings.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
D=/dev/shm/pgenhip_e2e_$$
mkdir -p $D
trap "rm -rf $D" EXIT
CLI=$ROOT/pgen_rs_amd/pgen-hip
V=${1:-1103547}; N=${2:-2504}
TIMEFORMAT='   wall %3R s (user %3U sys %3S)'
echo "== synth $V x $N"; time $CLI synth $D/chr22 --variants $V --samples $N || exit 1
ls -la $D | awk 'NR>1{print "  ", $5, $9}'
echo "== query all rows"; time $CLI query $D/chr22 -f 'ID' > /dev/null
echo "== filter keep-2 (README.md:164-168 shape)"; time $CLI filter $D/chr22 --include-var 'POS == "16647494" || POS == "16050007"' -o $D/two.vcf --stats
wc -c $D/two.vcf
echo "== filter keep-all (README.md:178-183 shape)"; time $CLI filter $D/chr22 -o $D/all.vcf --stats
wc -c $D/all.vcf; sha256sum $D/all.vcf | cut -c1-16
rm -f $D/all.vcf
echo "== filter keep-all -> BGZF (.vcf.gz), level 6 / level 1"; time $CLI filter $D/chr22 -o $D/all.vcf.gz --stats
wc -c $D/all.vcf.gz; rm -f $D/all.vcf.gz
time $CLI filter $D/chr22 -o $D/all.vcf.gz --bgzf-level 1 --stats
wc -c $D/all.vcf.gz; rm -f $D/all.vcf.gz
echo "== filter keep-all, 1% samples"; time $CLI filter $D/chr22 --include-sam 'KEEP == "1"' -o $D/k.vcf --stats
wc -c $D/k.vcf
