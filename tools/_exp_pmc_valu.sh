cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  rm -rf $R/gpurun_out/pv
  GPUNTT_LIB=$R/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/pv -o pmc -- python $R/tools/run_case.py c2 10 > /dev/null 2>&1
  echo $lib; cd $R; python tools/pmc_dump.py gpurun_out/pv | grep "INSTS_VALU" | grep -v prep | cut -c1-120; cd /tmp
  GPUNTT_LIB=$R/$lib python $R/tools/run_case.py c2 300 2>/dev/null | tail -1 | cut -c1-110
done
