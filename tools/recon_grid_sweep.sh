#!/bin/bash
# RECON persistent-grid sweep: one frame alone and the GOP-ordered stream
cd ${GRAFT_REPO_ROOT:-$PWD}
for g in 96 128 192 256 384; do
  export VVC355_RECON_GRID=$g
  a=$(timeout -k 10 200 python bench.py --gop 0 --frames-in-flight 1 --only intra_recon_wavefront --no-verify --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stages']['intra_recon_wavefront']['ms'],3))")
  echo "grid $g: RECON alone $a ms"
done
for g in 96 128 192 256 384; do
  export VVC355_RECON_GRID=$g
  b=$(timeout -k 10 400 python bench.py --no-verify --no-cpu-baseline --no-upload --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['independent_frames']['value'],1))")
  echo "grid $g: GOP-16 stream / independent frames/s $b"
done
