set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODEL=cvae-couple SHAPE=switchboard timeout -k 10 120 python3 $R/scripts/train_only.py lstm 100
MODEL=cvae-couple SHAPE=switchboard timeout -k 10 170 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_sw -o t -- python3 $R/scripts/train_only.py lstm 6 > $R/gpurun_out/tr_sw.log 2>&1
f=$(ls $R/gpurun_out/tr_sw/*/*kernel_trace.csv $R/gpurun_out/tr_sw/*kernel_trace.csv 2>/dev/null | head -n 1 || true)
python3 $R/scripts/trace_step.py $f > $R/gpurun_out/step_switchboard.txt
rm -rf $R/gpurun_out/tr_sw
