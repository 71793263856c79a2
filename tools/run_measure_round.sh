#!/bin/bash
# usage (build container): TAG=r5 tools/run_measure_round.sh [gpurun timeout seconds]
# The round's ONE full measurement set, as its last act: refuses while anything bench.py hashes (csrc/, include/) or bench.py itself
# differs from HEAD - the PMC stamp must be the round's last commit touching those files - records the commit in
# tools/_git_state (travels with the snapshot; tools/measure_round.sh refuses to run without it), runs tools/measure_round.sh on the
# GPU box and collects the summaries into profiles/.
set -e
cd "$(dirname "$0")/.."
TAG=${TAG:-r6}
PKG=wave-u-net-for-speech-enhancement_amd
if [ -n "$(git status --porcelain -- $PKG/csrc include bench.py tools/measure_round.sh tools/collect_round.py)" ]; then
    echo "refused: uncommitted changes under csrc/ include/ bench.py or the measurement tools:" >&2
    git status --porcelain -- $PKG/csrc include bench.py tools/measure_round.sh tools/collect_round.py >&2
    exit 2
fi
python -c "import __graft_entry__ as g; g.build()" > /dev/null
echo "$(git rev-parse HEAD) clean $(python -c 'import bench; print(bench.source_hash())')" > tools/_git_state
# PMC_ONLY=1 tools/run_measure_round.sh: the counter passes alone -> profiles/pmc_traffic.json (40 s of GPU time: re-stamp after any change under csrc/)
/usr/local/graft/bin/gpurun --timeout ${1:-2400} -- "TAG=$TAG PMC_ONLY=$PMC_ONLY bash tools/measure_round.sh"
python tools/collect_round.py $TAG ${PMC_ONLY:+--pmc-only}
