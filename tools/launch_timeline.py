"""host launch call against device execution, kernel by kernel, from a `rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d DIR` run:
   python tools/launch_timeline.py DIR [count] - for the last `count` kernels: when the host's launch call began / ended and when the kernel ran
   (microseconds relative to the first listed kernel's start), the idle time of the device in front of the kernel."""
import csv, glob, sys
d = sys.argv[1]; count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
af = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
api = {}
for r in csv.DictReader(open(af)):
    api[r["Correlation_Id"]] = (r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
rows = sorted(csv.DictReader(open(kf)), key=lambda r: int(r["Start_Timestamp"]))[-count:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = None
print("%-30s %10s %10s %10s %8s %8s" % ("kernel", "call@", "call_end@", "start@", "dur", "idle"))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = api.get(r["Correlation_Id"])
    name = r["Kernel_Name"].split("(")[0].replace("ppsfm::", "").replace("void ", "").replace(" ", "")[:30]
    print("%-30s %10.1f %10.1f %10.1f %8.2f %8.2f" % (name, (a[1] - t0) / 1e3 if a else float("nan"), (a[2] - t0) / 1e3 if a else float("nan"), (s - t0) / 1e3, (e - s) / 1e3,
                                                   (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
