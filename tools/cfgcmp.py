import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f)); t=d['linear_only_totals_one_gpu']
    print(f, {k:(v['ms_warm_requant'], v['ms_warm_cached']) for k,v in t.items()})
