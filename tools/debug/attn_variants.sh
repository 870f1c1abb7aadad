#!/bin/bash
# times the attention kernels with each measurement build tools/debug/variants/attn_*.so (ATTN_SKIP bits: flash_attn.hip)
cd $GRAFT_REPO_ROOT
echo "full: $(python tools/debug/attn_bwd_time.py 2>&1 | tail -1)"
for f in tools/debug/variants/attn_*.so; do
  echo "$(basename $f): $(MONOSOWA_ATTN_LIB=$PWD/$f timeout -k 5 100 python tools/debug/attn_bwd_time.py 2>&1 | tail -1)"
done
