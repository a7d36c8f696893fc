cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
for l in cmdlist graph eager; do python bench.py --no-cpu-baseline --launch $l 2>&1 | grep metric | cut -c1-200; done
for l in cmdlist graph eager; do GSAGE_FORCE_DDP=1 python bench.py --no-cpu-baseline --launch $l 2>&1 | grep metric | cut -c1-200; done
python tools/overlap_check.py 2>&1 | grep -E "exchange|Error|error"
