#!/bin/bash
# Build the REFERENCE's own Cython binding (cfast_slic) with the arch "hip/gfx950" added, linked against this repository's
# libfslic_hip.so: copy /root/reference to integration/_build/fast-slic, apply integration/hip_arch.patch (the hunks of
# INTEGRATION.md sections 2-3: cfast_slic.pxd, cfast_slic.pyx, fast_slic/hip.py, setup.py), run the reference's own setup.py.
# The result (integration/_build/fast-slic/{cfast_slic*.so, fast_slic/}) is git-ignored and travels to the GPU box like
# oracle/_ref; tests/test_gpu_ref_binding.py imports it from there.  Nothing from the reference enters the history.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
REF=${REF:-/root/reference}
if [ ! -d "$REF" ]; then echo "[integration] $REF absent: keeping the prebuilt integration/_build (if any)"; exit 0; fi
B=$HERE/_build
STAMP=$B/.stamp
NEW=$(cat "$HERE/hip_arch.patch" "$ROOT/include/fslic_hip.h" | sha1sum | cut -c1-16)
if [ -f "$STAMP" ] && [ "$(cat $STAMP)" = "$NEW" ] && ls $B/fast-slic/cfast_slic*.so >/dev/null 2>&1; then exit 0; fi
rm -rf "$B"; mkdir -p "$B"
cp -r "$REF" "$B/fast-slic"
rm -rf "$B/fast-slic/.git"
cd "$B/fast-slic"
patch -p1 -s < "$HERE/hip_arch.patch"
export FSLIC_HIP_INCLUDE=$ROOT/include
export FSLIC_HIP_LIBDIR=$ROOT/fast_slic_amd
export FSLIC_HIP_RPATH='$ORIGIN/../../../fast_slic_amd'
python3 setup.py -q build_ext --inplace > "$B/build.log" 2>&1 || { tail -30 "$B/build.log"; exit 1; }
rm -rf build
echo "$NEW" > "$STAMP"
echo "[integration] built $(ls $B/fast-slic/cfast_slic*.so)"
