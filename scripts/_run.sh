cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in libslamhip.so libslamhip_old.so; do
  rm -rf gpurun_out/prof_det
  SLAMHIP_LIB=$GRAFT_REPO_ROOT/slam.jl_amd/$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_det -o det -- python3 scripts/prof_detect.py 32 > /dev/null 2>&1
  echo $L; python - <<'PY'
import csv, glob, collections, statistics
f = glob.glob("gpurun_out/prof_det/**/*kernel_trace.csv", recursive=True)[0]
v=[(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "detect_cells" in r["Kernel_Name"]]
print(" ".join(f"{x:.0f}" for x in v[1::5]))
PY
done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
