"""Static guard of the 16-byte-store hazard workaround (rbis_kernels.hpp stg2): in a hipcc -S dump, no
instruction may write a data register of a `buffer_store_dwordx4` before the `s_nop` that follows the store.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o step.s pronto_amd/csrc/pb_step.hip
    python scripts/chk_store_hazard.py step.s         (exit code 1 on a violation; tests/test_isa_hazard.py runs it)"""
import re,sys
L=open(sys.argv[1]).read().split('\n')
bad=0; n=0; nonop=0
for i,l in enumerate(L):
    m=re.search(r'buffer_store_dwordx4 v\[(\d+):(\d+)\]',l)
    if not m: continue
    n+=1
    lo,hi=int(m.group(1)),int(m.group(2))
    j=i+1
    while j<len(L) and 's_nop' not in L[j] and 's_endpgm' not in L[j] and j-i<60:
        w=re.match(r'\s+(v_\w+|ds_read\w*|buffer_load\w*|global_load\w*|flat_load\w*|scratch_load\w*)\s+v\[?(\d+)(?::(\d+))?\]?',L[j])
        if w:
            a=int(w.group(2)); b=int(w.group(3) or a)
            if not (b<lo or a>hi): bad+=1; print("OVERLAP", l.strip(), "->", L[j].strip())
        j+=1
    if j-i>=60: nonop+=1
print("stores",n,"overwrites before nop",bad,"no nop within 60:",nonop)

sys.exit(1 if (bad or nonop) else 0)
