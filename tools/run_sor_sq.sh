# SQ / LDS counter passes over the flow series the bench runs (8 pairs of 1024^2, 512-thread SOR workgroups):
# which unit limits k_sor.  Separate --pmc passes with --kernel-trace only (the pool refuses other trace domains).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/sorsq_$tag -o b -- python tools/brox_pmc.py > gpurun_out/sorsq_$tag.log 2>&1 || echo fail_$tag
done
echo done
