set -o pipefail
export TMPDIR=/tmp MPI_OVERLAP=1 MPI_ITERS=4
rm -rf gpurun_out/prof_lanes && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_lanes -- python3 tools/mpi_profile.py > gpurun_out/r4_lanes_trace.txt 2>&1
cp $(find gpurun_out/prof_lanes -name "*kernel_trace.csv" | head -1) gpurun_out/r4_lanes_kernel_trace.csv
wc -l gpurun_out/r4_lanes_kernel_trace.csv
