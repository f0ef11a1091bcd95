#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r03_k1_stream.txt"
export TMPDIR=/tmp
cd "$R"
SDRM_K1_STREAM=1 timeout 1200 python -m pytest tests -m gpu -q -x -k "fixtures or lucky7 or ragged or mixed_rate or nan_inf or many_channel or batch_256 or random_config or long_symbols or doppler" 2>&1 | tail -2 | tee -a "$OUT"
SDRM_K1_STREAM=1 SDRM_K1_STREAM_TILES=1 timeout 1200 python -m pytest tests -m gpu -q -x -k "fixtures or ragged or mixed_rate" 2>&1 | tail -1 | tee -a "$OUT"
SDRM_K1_STREAM=1 python tools/stage_times.py 256 2>/dev/null | tail -1 | tee -a "$OUT"
python tools/stage_times.py 256 2>/dev/null | tail -1 | tee -a "$OUT"
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 2>/dev/null | tail -1); printf "  %-34s %s\n" "$label" "$r" | tee -a "$OUT"; }
n=131072
for ch in 256 1024 2048 4096; do
  echo "== chunk $n x $ch channels" | tee -a "$OUT"
  cell "tiled" SDRM_K1_STREAM=0
  cell "streaming" SDRM_K1_STREAM=1
  cell "tiled (again)" SDRM_K1_STREAM=0
  cell "streaming (again)" SDRM_K1_STREAM=1
  cell "streaming, 4 tiles per workgroup" SDRM_K1_STREAM=1 SDRM_K1_STREAM_TILES=4
  cell "streaming, 32 tiles per workgroup" SDRM_K1_STREAM=1 SDRM_K1_STREAM_TILES=32
done
