import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        for k in d:
            if k.startswith('roofline'):
                print('%-28s %.4f ms  %s' % (k, d[k].get('avg_launch_ms') or 0, d[k].get('kernel','')[:70]))
        if 'ms_per_step' in d: print('ms_per_step', d['ms_per_step'], d.get('value'))
