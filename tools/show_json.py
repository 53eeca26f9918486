import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f)); print(f)
    for k,v in d.items(): print("  ",k, v)
