cd /root/repo/tools
for v in "-DGSFM_LOOK_NT=1" "-DGSFM_LOOK_NT=2" "-DGSFM_LOOK_NT=3"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -o /tmp/bcb bench_chol_batch.hip 2>/dev/null
  echo "## $v"; for r in 1 2; do timeout 120 /tmp/bcb | grep -i "live\|one launch\|fused" | cut -c1-200; done
done
