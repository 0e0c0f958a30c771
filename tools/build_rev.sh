#!/bin/bash
# Build the kernels of another git revision as a variant library for A/B runs on ONE GPU box:
#   tools/build_rev.sh <rev> <tag>   ->  behindthescenes_amd/variants/libbts_<tag>.so   (load with BTS_RENDER_LIB / tools/lib_ab.py <tag>)
# Only csrc/ and include/ of <rev> are used; the host code stays the working tree's, so the C ABI must be compatible.
set -e
REV=$1; TAG=$2
REPO=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/bts_rev_$TAG
rm -rf $T && mkdir -p $T/behindthescenes_amd $T/tools
git -C $REPO archive $REV behindthescenes_amd/csrc include behindthescenes_amd/build.py tools/check_pk_opsel.py | tar -x -C $T
touch $T/behindthescenes_amd/__init__.py
(cd $T && BTS_OBJ_DIR=$T/obj python -c "
import sys; sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location('b', 'behindthescenes_amd/build.py'); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(b.build_library(force=True))")
mkdir -p $REPO/behindthescenes_amd/variants
cp $T/behindthescenes_amd/libbts_render.so $REPO/behindthescenes_amd/variants/libbts_$TAG.so
ls -la $REPO/behindthescenes_amd/variants/libbts_$TAG.so
