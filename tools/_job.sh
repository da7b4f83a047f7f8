cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/study; O=gpurun_out/study
python tools/tolerance_study.py --pairs 4541 > $O/cfg4.json 2> $O/cfg4.err
python tools/tolerance_study.py --pairs 128 --variant pca --mode direct7 --resolution 0.5 --azimuth 2048 > $O/cfg5_d7.json 2> $O/cfg5_d7.err
python tools/tolerance_study.py --pairs 128 --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 > $O/cfg5_d1.json 2> $O/cfg5_d1.err
python tools/tolerance_study.py --pairs 271 --variant pca --mode direct1 > $O/pca_d1.json 2> $O/pca_d1.err
python - <<PY
import json
for c in ("cfg4","cfg5_d7","cfg5_d1","pca_d1"):
    d=json.load(open("$O/%s.json"%c)); t=d["tolerance_vs_oracle"]
    print(c, {k:t[k] for k in ("pairs","iteration_flips","pairs_beyond_tolerance","pairs_flagged_by_the_engine","pairs_beyond_tolerance_and_not_flagged","pairs_beyond_tolerance_list","max_dtrans_m","p999_dtrans_m")})
PY
