# kernel timeline of ONE single-GPU proof (rocprofv3 --kernel-trace -> tools/trace_gaps.py): usage: bash tools/trace_single.sh <tag> "<bench flags>" [regex]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-s}; FLAGS=${2:---log-n 20 --table range}; PAT=${3:-.}
rm -rf $O/trs_$TAG
rocprofv3 --kernel-trace --output-format rocpd -d $O/trs_$TAG -- python3 bench.py $FLAGS --steps 3 --warmup 2 --no-cpu-baseline --no-inflight --no-profile > $O/trs_$TAG.log 2>&1
DB=$(find $O/trs_$TAG -name "*.db" | head -1)
cols=1; case "$FLAGS" in *"log-n 2"[2-9]*) cols=2;; esac; case "$FLAGS" in *and*|*xor*) cols=4;; esac; case "$FLAGS" in *"log-n"*) ;; *) cols=4;; esac
LH_TRACE_COLUMNS=$cols python3 tools/trace_gaps.py "$DB" "$PAT" > $O/gaps_single_$TAG.txt 2>&1
rm -rf $O/trs_$TAG
head -60 $O/gaps_single_$TAG.txt
