#!/bin/bash
# Build the REFERENCE's own Cython binding (cfast_slic) with the arch "hip/gfx950" added, linked against this repository's
# libfslic_hip.so: a scratch copy of /root/reference OUTSIDE this repository gets integration/hip_arch.patch (the hunks of
# INTEGRATION.md sections 2-3: cfast_slic.pxd, cfast_slic.pyx, fast_slic/hip.py, setup.py) and the reference's own setup.py
# builds it.  Only the built extension module comes back (integration/_build/cfast_slic*.so, git-ignored, travels to the GPU
# box like oracle/_ref); tests/test_gpu_ref_binding.py drives its SlicModel directly.  No reference source enters this
# repository, neither its history nor its working tree.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
REF=${REF:-/root/reference}
B=$HERE/_build
STAMP=$B/.stamp
NEW=$(cat "$HERE/hip_arch.patch" "$ROOT/include/fslic_hip.h" "$0" | sha1sum | cut -c1-16)
if [ ! -d "$REF" ]; then
  # no reference here (the GPU box): the prebuilt module is used only if it was built from THIS header and patch -- a module built
  # against another layout of fslic_params must not meet the new library
  if [ -f "$STAMP" ] && [ "$(cat $STAMP)" = "$NEW" ]; then echo "[integration] $REF absent: keeping the prebuilt integration/_build"
  elif [ -d "$B" ]; then echo "[integration] $REF absent and integration/_build is stale (header / patch changed since it was built): removed"; rm -rf "$B"
  fi
  exit 0
fi
if [ -f "$STAMP" ] && [ "$(cat $STAMP)" = "$NEW" ] && ls $B/cfast_slic*.so >/dev/null 2>&1; then exit 0; fi
rm -rf "$B"; mkdir -p "$B"
T=$(mktemp -d /tmp/fslic_refbuild.XXXXXX)
trap 'rm -rf "$T"' EXIT
cp -r "$REF" "$T/fast-slic"
rm -rf "$T/fast-slic/.git"
cd "$T/fast-slic"
patch -p1 -s < "$HERE/hip_arch.patch"
export FSLIC_HIP_INCLUDE=$ROOT/include
export FSLIC_HIP_LIBDIR=$ROOT/fast_slic_amd
export FSLIC_HIP_RPATH='$ORIGIN/../../fast_slic_amd'
python3 setup.py -q build_ext --inplace > "$B/build.log" 2>&1 || { tail -30 "$B/build.log"; exit 1; }
cp cfast_slic*.so "$B/"
# Only the compiled extension travels.  The reference's Python facade (fast_slic/*.py) stays where it is: neither source nor
# bytecode of it enters this repository or the GPU box.  Where /root/reference exists (this container) the facade hunk of the
# patch, fast_slic/hip.py, is imported against the scratch copy right here and the outcome kept in the build directory:
PYTHONPATH="$B:$T/fast-slic" python3 "$HERE/facade_check.py" > "$B/facade_import.log" 2>&1 || true
echo "$NEW" > "$STAMP"
echo "[integration] built $(ls $B/cfast_slic*.so)"
