import json,sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); pk=d["roofline"]["per_kernel"]
            print(f, d["value"], d["ms_per_step"], {k:(v["launches_per_step"],v["ms_per_launch"],v["tflops"]) for k,v in pk.items()})
