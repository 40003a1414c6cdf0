cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmcA -o b -- python tools/measure_split.py 3 > gpurun_out/pmcA.log 2>&1 || echo failA
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 --kernel-trace --output-format csv -d gpurun_out/pmcB -o b -- python tools/measure_split.py 3 > gpurun_out/pmcB.log 2>&1 || echo failB
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d gpurun_out/pmcC -o b -- python tools/measure_split.py 3 > gpurun_out/pmcC.log 2>&1 || echo failC
echo rc=0
