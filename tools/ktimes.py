import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
print(sys.argv[2], [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) // 1000 for r in rows])
