# build-service width: LANCET_SVC_WGS workgroups resident next to the batch's kernels
cd /root/repo
for n in ${@:-8 16 24 32 48 64}; do
  echo "== LANCET_SVC_WGS=$n"
  LANCET_SVC_WGS=$n timeout 120 python tools/quick_gpu.py bench 32768 2>&1 | grep -E "^run 2|LDS build kernel|service|total workgroup"
done
