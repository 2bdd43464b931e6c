import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print("value", d["value"], "ms", d["ms_per_step"])
st=d.get("stages",{})
print({k:round(v.get("avg_us",0),1) for k,v in st.items()})
