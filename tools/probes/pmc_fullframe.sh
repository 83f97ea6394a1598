# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of every kernel of the full-frame Farneback call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ffpmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/tools/time_flow_fullframe.py 25 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/tools/time_flow_fullframe.py 25 > $O/write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/fetch $O/write > $O/pmc_flow_fullframe.json
find $O -name "*counter_collection.csv" -size +20M -delete
head -c 1500 $O/pmc_flow_fullframe.json
