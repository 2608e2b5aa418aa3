cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export VOF2D_OVERLAP_HALVES=0 VOF2D_FUSE_TM=1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d /tmp/tmc -- python3 $R/tools/bound_run.py --steps 60 > /tmp/tmc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/tmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void vof::", "")[:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r: acc[k]["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, v in acc.items():
    if "k_tm" in k or "k_momentum" in k or "k_transport" in k or "jacobi_tb" in k:
        m = {c: sum(x) / len(x) for c, x in v.items()}
        cells = 4096 * 4096
        print("%-40s n %3d | VALU/cell %.2f SALU/cell %.2f LDS/cell %.3f | wait_any %.0f%% wait_inst %.0f%% active %.0f%% of wave cycles | dur(pmc) %.0f us | VALU busy (4.4 cy) %.0f%%" % (
            k, len(v["SQ_INSTS_VALU"]), m["SQ_INSTS_VALU"] / cells, m["SQ_INSTS_SALU"] / cells, m.get("SQ_INSTS_LDS", 0) / cells,
            100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"],
            m.get("dur", 0) / 1e3, 100 * m["SQ_INSTS_VALU"] * 4.4 / (1024 * m.get("dur", 1) * 2.4) if m.get("dur") else 0))
PY
