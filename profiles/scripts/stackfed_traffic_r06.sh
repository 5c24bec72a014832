#!/bin/bash
# Round 6: HBM traffic (FETCH_SIZE, WRITE_SIZE: separate --pmc passes) and instruction counters of ONE stack-fed sweep launch, per shape, form and
# stack layout -- the cooperative nx = 12 sweep on both layouts, the one-lane C2 shape at its chip-filling batch.  Output: gpurun_out/sf_traffic/*.md
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/sf_traffic; mkdir -p $O; rm -rf $O/*
run() {   # run <tag> <layout> <nx nu m N B form>
  local tag=$1 lay=$2; shift 2
  local i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    CDDP_HIP_STACKS_LAYOUT=$lay timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${tag}_$i -o r -- python profiles/scripts/stackfed_one.py "$@" > $O/pmc_${tag}_$i.log 2>&1
  done
  python profiles/summarize_pmc.py $O/pmc_${tag}_* > $O/counters_$tag.md
  tail -2 $O/pmc_${tag}_1.log
  rm -rf $O/pmc_${tag}_*/
}
run nx12_path_4096_coop_t4 t4 12 4 8 400 4096 coop
run nx12_path_4096_coop_plain plain 12 4 8 400 4096 coop
run nx12_clddp_4096_coop_t4 t4 12 4 0 400 4096 coop
run nx14_path_3072_coop_t4 t4 14 7 14 150 3072 coop
run nx14_clddp_4096_coop_t4 t4 14 7 0 150 4096 coop
run nx4_path_131072_lane_plain plain 4 1 2 100 131072 lane
run nx4_clddp_65536_lane_plain plain 4 1 0 100 65536 lane
run nx3_path_65536_lane_plain plain 3 2 5 200 65536 lane
run nx3_clddp_131072_lane_plain plain 3 2 0 200 131072 lane
python - <<'PY'
import glob, json, os, re
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on profiles/scripts/stackfed_one.py, one sweep launch per dispatch (profiles/scripts/stackfed_traffic_r06.sh)",
       "unit": "bytes per launch = FETCH_SIZE x 1024 x 2 (gfx950 correction, profiles/r01_c_pmc_calibration.md) + WRITE_SIZE x 1024", "launches": {}}
for f in sorted(glob.glob("gpurun_out/sf_traffic/counters_*.md")):
    tag = os.path.basename(f)[9:-3]
    rows = [l.strip().strip("|").split("|") for l in open(f) if l.startswith("|")]
    hdr = [c.strip() for c in rows[0]]
    for r in rows[2:]:
        c = [x.strip() for x in r]
        if c[0].strip("`").strip() in ("void", "void ") or "k_stacks_backward" in c[0]:
            d = dict(zip(hdr, c))
            fe, wr = float(d["FETCH_SIZE"]), float(d["WRITE_SIZE"])
            out["launches"][tag] = {"fetch_kb": fe, "write_kb": wr, "bytes_per_launch": fe * 2048 + wr * 1024,
                                    "vmem_rd_insts": float(d.get("SQ_INSTS_VMEM_RD") or 0), "vmem_wr_insts": float(d.get("SQ_INSTS_VMEM_WR") or 0),
                                    "valu_insts": float(d.get("SQ_INSTS_VALU") or 0), "lds_insts": float(d.get("SQ_INSTS_LDS") or 0)}
json.dump(out, open("gpurun_out/sf_traffic/r06_pmc_traffic_stackfed.json", "w"), indent=1)
for k, v in out["launches"].items(): print(k, "%.3f GB" % (v["bytes_per_launch"] / 1e9))
PY
