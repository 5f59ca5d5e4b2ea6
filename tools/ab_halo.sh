# A/B of two builds of libsdhip on ONE box, interleaved (cdna guide rule 24): SD_AMD_LIB selects the library.
# usage: tools/ab_halo.sh <other libsdhip.so> [bench_ops --only arg]
other=$1; only=${2:-conv}
for i in 1 2; do
  echo "== default build"; python tools/bench_ops.py --only $only 2>&1 | grep -v "amdgpu.ids" | tail -13
  echo "== $other"; SD_AMD_LIB=$other python tools/bench_ops.py --only $only 2>&1 | grep -v "amdgpu.ids" | tail -13
done
