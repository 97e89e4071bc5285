#!/bin/sh
# Build and run the standalone packed-FP32 co-run repro (tools/exp/pk_corun.hip): two binaries — the victim kernel with and
# without v_pk_*_f32 — plus the ISA of rbf_aggregate_bwd_kernel of both (profiles/r5_pk_isa_*.s are the committed copies).
#   sh tools/exp/pk_corun.sh [out_dir = gpurun_out/pk_corun] [replays = 200]        (from the repo root or from tools/exp)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=${1:-$ROOT/gpurun_out/pk_corun}
REPS=${2:-200}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CSRC=$ROOT/gemnet_pytorch_amd/csrc
mkdir -p "$OUT" "$HERE/bin"
SRCS="$HERE/pk_corun.hip $CSRC/aggregate.hip $CSRC/chain2.hip $CSRC/chain3.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -I $ROOT/include"
# the failing form is the ROUND-2 adjoint kernel (-DGN_AGG_V1) with packed FP32 (-DGN_AGG_PK); pk2 = the round-5 kernel with
# packed FP32 (is the restructured kernel a victim too?); the product build has neither (no-packed-fp32-ops file-wide)
for v in pk nopk pk2; do
  D="-DGN_AGG_V1"; [ $v = pk ] && D="-DGN_AGG_V1 -DGN_AGG_PK"; [ $v = pk2 ] && D="-DGN_AGG_PK"
  if [ ! -x "$HERE/bin/pk_corun_$v" ] || [ "$HERE/pk_corun.hip" -nt "$HERE/bin/pk_corun_$v" ]; then
    $HIPCC $FLAGS $D $SRCS -o "$HERE/bin/pk_corun_$v"
  fi
  # ISA of the victim kernel alone
  $HIPCC $FLAGS $D -S --cuda-device-only "$CSRC/aggregate.hip" -o "$OUT/aggregate_$v.s" 2>/dev/null
  K=rbf_aggregate_bwd_kernelE; [ $v = pk2 ] && K=rbf_aggregate_bwd_kernel_v2E
  awk -v k="$K" '$0 ~ "^_ZN[^ ]*" k "[^ ]*:" {on=1} on {print} /s_endpgm/ {if (on) exit}' "$OUT/aggregate_$v.s" > "$OUT/rbf_aggregate_bwd_$v.s"
  echo "$v: $(grep -c 'v_pk_.*_f32' "$OUT/rbf_aggregate_bwd_$v.s" || true) v_pk_*_f32 instructions in rbf_aggregate_bwd_kernel"
done
if [ -e /dev/kfd ]; then
  for v in pk nopk pk2; do
    for br in 2 1; do
      echo "== pk_corun_$v $REPS $br"
      "$HERE/bin/pk_corun_$v" "$REPS" "$br" 2>&1 | tee "$OUT/run_${v}_${br}branch.txt"
    done
  done
else
  echo "(no GPU here: binaries built, ISA dumped; run this script on the gfx950 box)"
fi
