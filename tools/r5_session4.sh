#!/bin/bash
# Round-5 fourth GPU session: the tie-rule option -- its GPU test first, then the whole suite (the streams changed by three
# prologue instructions), a bench line
set -e
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "tfhe_rs or server_mode" > $O/${tag}_pytest_new.log 2>&1 || { tail -60 $O/${tag}_pytest_new.log; exit 1; }
tail -3 $O/${tag}_pytest_new.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/${tag}_pytest.log 2>&1 || { tail -40 $O/${tag}_pytest.log; exit 1; }
tail -3 $O/${tag}_pytest.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/${tag}_bench.log 2>&1 || { tail -20 $O/${tag}_bench.log; exit 1; }
tail -1 $O/${tag}_bench.log > $O/${tag}_bench_line.json
python -c "
import json; l=json.load(open('$O/${tag}_bench_line.json')); print('value', l['value'], 'host result', l['value_host_result'], 'stage_s', l['stage_s']['aes_decryption'], l['stage_s']['compute'])"
