#!/bin/bash
# Installs the MI355X drop-in into a checkout of SNUCP/MKHE-KKLSS (a machine with Go, lattigo v2.3.0 in the module cache and ROCm):
#     shim/go/install.sh /path/to/MKHE-KKLSS [--gpu-evaluator]
# 1. the cgo binding as package mk-lattigo/mkrlwegpu (a leaf: it imports lattigo only);
# 2. the drop-in types as files of the reference's own packages, behind the build tag mkhe_gpu (mkrlwe.KeySwitcher replaced; mkckks / mkbfv
#    GPUEvaluator added; *_gpu_off.go keep the names defined without the tag);
# 3. `//go:build !mkhe_gpu` on the two reference files the KeySwitcher replaces (patches/mkrlwe_build_tags.diff);
# 4. with --gpu-evaluator: the reference's mkckks / mkbfv tests construct the GPUEvaluator (two lines each).
# Then:  CGO_CFLAGS=-I<this repo>/include CGO_LDFLAGS="-L<this repo>/mkhe-kklss_amd/lib -lmkhe_hip" LD_LIBRARY_PATH=<this repo>/mkhe-kklss_amd/lib \
#        go test -tags mkhe_gpu ./mkrlwe ./mkckks ./mkbfv            (without -tags: the untouched pure-Go reference)
set -euo pipefail
REF=${1:?usage: install.sh /path/to/MKHE-KKLSS [--gpu-evaluator]}
HERE=$(cd "$(dirname "$0")" && pwd)
[ -f "$REF/go.mod" ] && grep -q '^module mk-lattigo' "$REF/go.mod" || { echo "$REF is not a checkout of the reference (module mk-lattigo)"; exit 1; }
mkdir -p "$REF/mkrlwegpu"
cp "$HERE"/mkrlwegpu/*.go "$REF/mkrlwegpu/"
for pkg in mkrlwe mkckks mkbfv; do cp "$HERE"/dropin/$pkg/*.go "$REF/$pkg/"; done
patch -d "$REF" -p1 --forward < "$HERE/patches/mkrlwe_build_tags.diff"
if [ "${2:-}" = "--gpu-evaluator" ]; then
    patch -d "$REF" -p1 --forward < "$HERE/patches/mkckks_tests_gpu_evaluator.diff"
    patch -d "$REF" -p1 --forward < "$HERE/patches/mkbfv_tests_gpu_evaluator.diff"
fi
echo "installed; build with -tags mkhe_gpu"
