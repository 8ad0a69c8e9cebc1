# step rate under a few HIP runtime environment knobs (tools/exp/README.md)
for kv in X=0 AMD_OPT_FLUSH=0 ROC_SYSTEM_SCOPE_SIGNAL=0 DEBUG_HIP_KERNARG_COPY_OPT=0 ROC_USE_FGS_KERNARG=0 GPU_MAX_HW_QUEUES=1 ROC_AQL_QUEUE_SIZE=16384 HSA_ENABLE_INTERRUPT=0 ROC_SKIP_KERNEL_ARG_COPY=1; do
  echo -n "$kv  "; env $kv REPS=2 STEPS=4000 timeout 100 python tools/exp/rate.py 2>&1 | tail -1
done
