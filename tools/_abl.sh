cd $GRAFT_REPO_ROOT
for v in default cabl1 cabl2 cabl4 cabl7; do
  if [ $v = default ]; then L=$PWD/behindthescenes_amd/libbts_render.so; else L=$PWD/behindthescenes_amd/variants/libbts_$v.so; fi
  echo "== $v"; BTS_RENDER_LIB=$L BTS_ALLOW_LIB_OVERRIDE=1 python tools/conv_probe.py 5 2>&1 | grep -v amdgpu.ids
done
