#!/bin/bash
# Rebuilds tools/ab_libs/libtftfund_A.so -- the library of an earlier commit (default: a9d655e, the end of round 4), built unchanged -- as the
# baseline of the same-box A/B runs (tools/ab_libs.py, tools/ab_libs_config4.py; profiles/r5_ab_libs.txt).  Usage: bash tools/build_ab_baseline.sh [commit] [tag]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
COMMIT=${1:-a9d655e}; TAG=${2:-A}
TMP=$(mktemp -d)
git -C "$ROOT" archive "$COMMIT" tft_vs_fund_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/tools/ab_libs"
( cd "$TMP/tft_vs_fund_amd/csrc" && ${HIPCC:-/opt/rocm/bin/hipcc} -O3 --offload-arch=gfx950 -std=c++17 -I. -shared -fPIC -o "$ROOT/tools/ab_libs/libtftfund_$TAG.so" capi.hip )
rm -rf "$TMP"
echo "$ROOT/tools/ab_libs/libtftfund_$TAG.so"
