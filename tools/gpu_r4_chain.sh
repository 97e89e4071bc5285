#!/bin/bash
# Round-4 GPU session: the wide chain layout (csrc/chain3.hip) — parity tests, per-program timings, bench A/B.
OUT=gpurun_out/${1:-r4_chain}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -1 $OUT/build.log
echo "== kernel tests"; timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "chain or two_plane or wide" > $OUT/pytest_chain.log 2>&1; tail -4 $OUT/pytest_chain.log
echo "== programs"; TILE_ROWS=${TILE_ROWS:-32,40,48} PYTHONPATH=. timeout 600 python tools/chain_programs.py > $OUT/programs.txt 2> $OUT/programs.err; cat $OUT/programs.txt
echo "== bench wide"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_wide.json 2> $OUT/bench_wide.log; cut -c1-260 $OUT/bench_wide.json
echo "== bench tall"; GEMNET_CHAIN_LAYOUT=tall timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_tall.json 2> $OUT/bench_tall.log; cut -c1-260 $OUT/bench_tall.json
