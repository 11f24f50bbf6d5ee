#!/usr/bin/env python3
"""Print the synchronisation skeleton (waits, barriers, LDS-DMA, branches, MFMA/LDS counts between them) of one kernel in a
hipcc -save-temps .s file.  usage: isa_loop.py file.s 'regex on mangled name'"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n')
pat = re.compile(sys.argv[2])
start = [i for i, l in enumerate(L) if l.endswith(tuple([':'])) is False and re.match(r'^_Z\S+:', l) and pat.search(l)][0]
end = next(i for i in range(start, len(L)) if 's_endpgm' in L[i])
cnt = collections.Counter()
def flush():
    global cnt
    if cnt: print('      ', dict(cnt))
    cnt = collections.Counter()
for l in L[start:end]:
    st = l.strip()
    op = re.split(r'[ \t]', st)[0]
    if op.startswith('s_waitcnt') or op.startswith('s_barrier') or 'global_load_lds' in op or op.startswith('s_cbranch') or re.match(r'^\.LBB', st):
        flush(); print(st[:90])
    elif op.startswith(('v_mfma', 'ds_read', 'ds_write', 'global_load', 'global_store', 'v_cvt_pk')):
        cnt[op.split('_e')[0][:24]] += 1
flush()
