#!/bin/bash
# Soak of the PCSHELL driver's process lifetime (one glibc "double free or corruption" abort of
# tests/cpp/build/petsc_driver in ~150 runs on a GPU box): N fresh processes, cycling through the tree-walk /
# vector-type / operator-type variants of tests/test_gpu_petsc.py, under glibc's heap checker (MALLOC_CHECK_=3,
# MALLOC_PERTURB_), every one required to print its complete report and exit 0.
# usage: soak_petsc_driver.sh [N=1000] [out=gpurun_out/r03_soak_petsc.txt]
N=${1:-1000}
OUT=${2:-gpurun_out/r03_soak_petsc.txt}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/tests/cpp/build/petsc_driver
G=$ROOT/tests/golden
mkdir -p "$(dirname "$OUT")"
TMP=$(mktemp -d)
COMMON="-blasted_async_fact_init_type init_original -blasted_async_apply_init_type init_zero -blasted_thread_chunk_size 128 -blasted_use_symmetric_scaling 0"
VARIANTS=(
 "baij seq -pc_type ksp -sub_pc_type shell -blasted_pc_type seqilu0 -blasted_async_sweeps 1,1"
 "baij seq -pc_type asm -sub_pc_type shell -blasted_pc_type seqilu0 -blasted_async_sweeps 1,1"
 "baij seq -pc_type shell -blasted_pc_type seqilu0 -blasted_async_sweeps 1,1"
 "baij hip -pc_type bjacobi -sub_pc_type shell -blasted_pc_type ilu0 -blasted_async_sweeps 3,3"
 "aij seq -pc_type bjacobi -sub_pc_type shell -blasted_pc_type sgs -blasted_async_sweeps 1,3"
 "baij seq -pc_type bjacobi -sub_pc_type shell -blasted_pc_type ilu0 -blasted_async_sweeps 3,3 -blasted_pin_host_arrays 1"
 "baij hip -pc_type ksp -sub_pc_type shell -blasted_pc_type level_sgs -blasted_async_sweeps 1,1"
 "aij seq -pc_type ksp -sub_pc_type shell -blasted_pc_type async_level_ilu0 -blasted_async_sweeps 4,1"
)
bad=0
t0=$(date +%s)
echo "# soak of $D: $N processes, MALLOC_CHECK_=3 MALLOC_PERTURB_=165, ${#VARIANTS[@]} variants in rotation" > "$OUT"
for ((i = 0; i < N; i++)); do
  v=(${VARIANTS[$((i % ${#VARIANTS[@]}))]})
  MALLOC_CHECK_=3 MALLOC_PERTURB_=165 "$D" --mat_file $G/2dcyl1.pmat --mat_type ${v[0]} --vec_type ${v[1]} --out $TMP/o -- ${v[@]:2} $COMMON > $TMP/out.txt 2> $TMP/err.txt
  rc=$?
  if [ $rc -ne 0 ] || ! grep -q "^done = 1" $TMP/out.txt; then
    bad=$((bad + 1))
    { echo "run $i (variant $((i % ${#VARIANTS[@]}))): rc=$rc"; echo "--- stdout tail"; tail -5 $TMP/out.txt; echo "--- stderr tail"; tail -20 $TMP/err.txt; } >> "$OUT"
  fi
  if [ $(((i + 1) % 100)) -eq 0 ]; then
    echo "$((i + 1)) runs, $bad not clean, $(( $(date +%s) - t0 )) s" | tee -a "$OUT"
  fi
done
echo "total: $N runs, $bad not clean" | tee -a "$OUT"
rm -rf $TMP
[ $bad -eq 0 ]
