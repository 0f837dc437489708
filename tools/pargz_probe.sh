#!/bin/bash
# host-side timing of the CLI's gzip readers on the e2e bench's files (after tools/e2e_bench.py --gz wrote them): gzread against cm_pargz.h
cd $GRAFT_REPO_ROOT
D=${1:-/tmp/chromap_amd_e2e}
F=$(ls $D/*_1.fq.gz 2>/dev/null | head -1)
[ -z "$F" ] && F=$(ls $D/*.gz | grep -v bgz | head -1)
echo "file $F $(stat -c %s $F) bytes"
t() { local s=$(date +%s%N); "$@" > /dev/null 2> /tmp/pargz_err.txt; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms  $(tail -1 /tmp/pargz_err.txt)"; }
echo -n "gzread: "; CM_PARGZ=0 t ./chromap_amd/chromap-amd --inflate-only $F
for T in 4 8 16 32 64; do echo -n "pargz $T threads: "; CM_PARGZ_THREADS=$T t ./chromap_amd/chromap-amd --inflate-only $F; done
