# A/B of libvlg builds inside ONE gpurun call (box-to-box differences are several per cent): tools/ab/libvlg_<tag>.so
# usage: bash tools/ab/run_ab.sh <only-filter> tag1 tag2 ...   ("cur" = the in-tree build)
only=$1; shift
for t in "$@"; do
  if [ "$t" = cur ]; then unset VLG_HIP_LIB; else export VLG_HIP_LIB=$PWD/tools/ab/libvlg_$t.so; fi
  python tools/kernel_bench.py --rounds 5 --only "$only" 2>/dev/null | cut -c1-46 > gpurun_out/ab_$t.txt || exit 1
done
f=gpurun_out/ab_$1.txt; shift
for t in "$@"; do paste -d'|' $f <(cut -c33-46 gpurun_out/ab_$t.txt) > gpurun_out/ab_tmp.txt; cp gpurun_out/ab_tmp.txt gpurun_out/ab_join.txt; f=gpurun_out/ab_join.txt; done
cat $f
