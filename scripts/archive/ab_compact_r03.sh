#!/bin/bash
# Round 3: compact-record kernel, frames per workgroup x next-frame prefetch, bench regime (back-to-back launches).
cd "$GRAFT_REPO_ROOT"
export AB_COMPACT=1 AB_SET=prefetch AB_NOSYNC=1 AB_ROUNDS=30
python3 scripts/ab_scan.py 1080p_dense8x8 4k_dense8x8
AB_FRAMES=1024 python3 scripts/ab_scan.py 4k_fine
echo "--- larger batches"
AB_FRAMES=16384 python3 scripts/ab_scan.py 1080p_dense8x8
AB_FRAMES=4096 python3 scripts/ab_scan.py 4k_dense8x8
echo "--- 40-byte records, same knobs (must not regress)"
AB_COMPACT=0 python3 scripts/ab_scan.py 1080p_dense8x8 4k_dense8x8
