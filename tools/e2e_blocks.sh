ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
D=/dev/shm/pgenhip_e2eb_$$
mkdir -p $D; trap "rm -rf $D" EXIT
CLI=$ROOT/pgen_rs_amd/pgen-hip
$CLI synth $D/c --variants 1103547 --samples 2504 || exit 1
TIMEFORMAT='   wall %3R s (user %3U sys %3S)'
for mib in 512 256 128 64; do
  echo "== block-mib $mib keep-all"; time $CLI filter $D/c -o $D/all.vcf --block-mib $mib --stats | cut -c100-; rm -f $D/all.vcf
  echo "== block-mib $mib 1% samples"; time $CLI filter $D/c --include-sam 'KEEP == "1"' -o $D/k.vcf --block-mib $mib --stats | cut -c100-; rm -f $D/k.vcf
done
