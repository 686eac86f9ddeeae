import csv,glob
f=glob.glob("gpurun_out/prof_members/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("k_members","k_point_clusters","k_cluster_heads","rocprim","k_segment","k_frame_cluster")): print(n.replace("void hope::","")[:60], r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3))
