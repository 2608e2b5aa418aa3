# Same-box A/B of library variants plus their instruction counters (one gpurun call):
#   bash tools/quick_ab.sh "<variant_ab args>" name1 name2 ...     e.g.  bash tools/quick_ab.sh "--reps 2 --steps 400" base t1
# -> gpurun_out/qab_<names>.log, gpurun_out/qab_<name>_bound_insts.json
A="$1"; shift
cd $GRAFT_REPO_ROOT
tag=$(echo "$*" | tr ' ' '_')
python3 tools/variant_ab.py $A "$@" > gpurun_out/qab_$tag.log 2>&1
cat gpurun_out/qab_$tag.log
for v in "$@"; do
  lib=taichi-2d-vof_amd/csrc/build/variants/libvof2d_$v.so
  [ "$v" = base ] && lib=taichi-2d-vof_amd/csrc/build/libvof2d_hip.so
  PASSES="insts" bash tools/collect_bound.sh qab_$v --steps 60 --lib $lib $QAB_RUN_ARGS > /dev/null 2>&1
  python3 - <<PY
import json
r=json.load(open("gpurun_out/qab_${v}_bound_insts.json"))
w=r["windows"]["11-60"]
for k,c in sorted(w.items()):
    print("%-8s %-22s VALU/cell %.3f SALU/cell %.3f branch %.3f vmem %.3f  us(pmc) %.1f" % ("$v", k, c.get("SQ_INSTS_VALU",0)/(${QAB_CELLS:-16777216}), c.get("SQ_INSTS_SALU",0)/(${QAB_CELLS:-16777216}), c.get("SQ_INSTS_BRANCH",0)/(${QAB_CELLS:-16777216}), (c.get("SQ_INSTS_VMEM_RD",0)+c.get("SQ_INSTS_VMEM_WR",0))/(${QAB_CELLS:-16777216}), c["_us"]))
PY
done
