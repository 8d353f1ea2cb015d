import json, csv, re, sys, glob
P='/root/repo/profiles/'
def J(f): return json.load(open(P+f))
d=J('r03_bench.json'); r=d['roofline']; h=d['host_inputs_leg']; c=d['cpu_baseline']; s=c['legs']['serving_shape']
print('BENCH', d['value'], d['latency_ms']['p50'], d['latency_ms']['p99'], 'ms/step', d['ms_per_step'])
print(' gather', r['avg_launch_us'], r['achieved'], r['frac'], 'alone', r['gather_alone']['avg_launch_us'], r['gather_alone']['frac'], 'single', r['single_query_launch']['avg_launch_us'], r['single_query_launch']['frac'], 'traffic', r['traffic'], 'sust', r['sustained_over_timed_region']['frac'], 'mlp', r.get('mlp_end_to_end',{}).get('achieved'))
print(' host', h['value'], h['h2d_GBps'], 'sets', h['launch_sets']['value'], h['launch_sets']['h2d_GBps'])
print(' cpu torch', c['value'], 'port', c['legs']['oracle_port']['value'], 'serving', s['value'], s['at']['p99_ms'], 'sat', s['saturation_closed_loop']['qps'])
for f in sorted(glob.glob(P+'r03_*bench*.json'))+sorted(glob.glob(P+'r03_*traced.json')):
    try:
        x=json.load(open(f)); rr=x['roofline']; m=rr.get('mlp_end_to_end') or {}
        cb=(x.get('cpu_baseline') or {}).get('value')
        print('%-40s %9.1f p99 %.3f gather %.2f us %.4f alone %s mlp %s %s cpu %s' % (f.split('/')[-1], x['value'], x['latency_ms']['p99'], rr['avg_launch_us'], rr['frac'], (rr.get('gather_alone') or {}).get('frac'), m.get('achieved'), m.get('frac'), cb))
    except Exception as e: print(f, 'ERR', e)
print(open(P+'r03_kernel_trace_by_grid.txt').read().split('\n')[1][60:140])
print(open(P+'r03_kernel_trace_by_grid.txt').read().split('\n')[2][60:140])
print(open(P+'r03_kernel_timeline.txt').read().strip().split('\n')[-2:])
for f in ('4','3','2'):
    for row in csv.DictReader(open(P+'r03_mlp_alone_stream%s_rocprofv3_kernel_stats.csv'%f)):
        if 'stream' in row['Name'] and 'pack' not in row['Name']: print('mlp alone', f, row['Name'][30:70], row['AverageNs'])
def pmc(f, pat):
    out={}
    for l in open(P+f):
        m=re.match(r'(\S.*?)\s+grid=(\d+)\s+(\S+)\s+n=\d+\s+avg=([\d.]+)', l)
        if m and re.search(pat, m.group(1)): out[(m.group(1).strip(), m.group(2), m.group(3))]=float(m.group(4))
    return out
for f,pat in (('r03_wnd_pmc_summary.txt','gemm|stream'),('r03_rmc3_pmc_summary.txt','gemm|stream|sls'),('r03_dien_pmc_summary.txt','rnn'),('r03_mlp_alone_pmc_summary.txt','stream4'),('r03_pmc_summary.txt','sls_flatc'),('r03_din_pmc_summary.txt','din_fused')):
    o=pmc(f,pat)
    for (k,g,cn),v in sorted(o.items()):
        if cn in ('SQ_VALU_MFMA_BUSY_CYCLES','GRBM_GUI_ACTIVE','FETCH_SIZE','WRITE_SIZE','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_WAVE_CYCLES'): print(f[4:12], k[:34], g, cn, v)
print(open(P+'r03_rmc3_single_stream_kernel_trace_by_grid.txt').read()[:700])
for l in open(P+'r03_traces_gather.jsonl'):
    x=json.loads(l); print('trace', x['profile'], x['queries_per_s'], x['gather_frac_of_8TBps'])
for f in ('r03_serve_rmc1.json','r03_serve_mix_wnd_ncf.json'): print(f, round(J(f)['qps']))
print(open(P+'r03_pcie_probe.txt').read()[-300:])
print(open(P+'r03_mlp_timeline_stream4.txt').read().strip().split('\n')[-1][:700])
