DCL_LINEAR_F16X3_ROWS=262144 bash tools/profile_config4.sh ab1 > /dev/null 2>&1
bash tools/profile_config4.sh ab0 > /dev/null 2>&1
head -1 gpurun_out/ab1_config4_kernels.csv | cut -c1-200; head -1 gpurun_out/ab0_config4_kernels.csv | cut -c1-200
