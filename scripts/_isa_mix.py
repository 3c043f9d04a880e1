import re,collections,sys
s=open(sys.argv[1]).read()
for name in sys.argv[2:]:
    i=s.find('\n_ZN2pb13'+name)
    j=s.find(':\n',i); e=s.find('s_endpgm',j)
    body=s[j:e]
    cnt=collections.Counter()
    for ln in body.splitlines():
        ln=ln.strip()
        if not ln or ln.startswith(';') or ln.startswith('.') or ln.endswith(':'): continue
        op=ln.split()[0]
        key=op
        if op.startswith(('v_fma_f64','v_mul_f64','v_add_f64','v_fmac_f64')): key='v_f64_arith'
        elif op.startswith(('ds_read','ds_load')): key='ds_read'
        elif op.startswith(('ds_write','ds_store')): key='ds_write'
        elif op.startswith('scratch_load'): key='scratch_load'
        elif op.startswith('scratch_store'): key='scratch_store'
        elif op.startswith('global_load'): key='global_load'
        elif op.startswith('global_store'): key='global_store'
        elif op.startswith('s_load'): key='s_load'
        elif op.startswith('s_waitcnt'): key='s_waitcnt'
        elif op.startswith(('v_mov','v_accvgpr')): key='v_mov/acc'
        elif op.startswith('v_'): key='v_other'
        elif op.startswith('s_'): key='s_other'
        cnt[key]+=1
    print(name, sum(cnt.values()), dict(cnt.most_common()))
