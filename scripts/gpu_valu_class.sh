# round 4: calibrate the VALU class counters and SQ_ACTIVE_INST_VALU on the opcode probe (input of scripts/limiters_json.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/valu_class; rm -rf $OUT; mkdir -p $OUT
P=$GRAFT_REPO_ROOT/scripts/probes/_bin/valu_rate_probe
[ -x $P ] || hipcc --offload-arch=gfx950 -O3 -o $P $GRAFT_REPO_ROOT/scripts/probes/valu_rate_probe.hip 2>/dev/null
cd /tmp
$P > $OUT/probe_stdout.txt 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT -d $OUT/p1 -o p1 -- $P > $OUT/p1.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_FMA_F16 -d $OUT/p2 -o p2 -- $P > $OUT/p2.log 2>&1
python $GRAFT_REPO_ROOT/scripts/valu_class_calibration.py $OUT/probe_stdout.txt $OUT/valu_classes.json $OUT/p1/p1_results.db $OUT/p2/p2_results.db > $OUT/calibration.txt 2>&1
rm -rf $OUT/*/*.db
tail -70 $OUT/calibration.txt
