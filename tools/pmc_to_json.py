"""Folds the FETCH_SIZE / WRITE_SIZE summaries written by tools/pmc_run.sh into profiles/pmc_traffic.json (read by bench.py).
usage: python tools/pmc_to_json.py <workload key, e.g. REBLUR_DIFFUSE_SPECULAR_2560x1440> <fetch summary.txt> <write summary.txt> <source note> [<SQ summary.txt>]
The optional SQ summary (the counter set with SQ_INSTS_VALU and SQ_WAVES) adds the executed VALU instructions per wave and the waves per launch."""
import json
import os
import re
import sys

KERNEL_TO_SHADER = [  # kernel-name fragments -> shader file name suffix (family prefix is added from the workload)
    ("ClassifyTiles", "ClassifyTiles.cs"), ("TemporalAccumulation", "TemporalAccumulation.cs"), ("HistoryFix", "HistoryFix.cs"), ("HistoryClamping", "HistoryClamping.cs"),
    ("TemporalStabilization", "TemporalStabilization.cs"), ("AtrousSmem", "AtrousSmem.cs"), ("RelaxAtrousKernel", "Atrous.cs"), ("PrePass", "PrePass.cs"),
    ("SpatialMode)0", "PrePass.cs"), ("SpatialMode)1", "Blur.cs"), ("SpatialMode)2", "PostBlur.cs"),
]


def library_digest():
    """the digest of the library the counters were taken from (raytracingdenoiser_amd/build.py writes it next to the .so: sources + headers + flags). bench.py compares it with
    the library it times and marks the recorded fields stale when they differ (the r05 driver line quoted counters from a library one kernel change older)."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracingdenoiser_amd", "lib", "libNRD_hip.so.digest")
    return open(path).read().strip() if os.path.exists(path) else None


def parse(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.e+-]+)\s*$", line)
        if m:
            out[m.group(1).strip()] = float(m.group(4))
    return out


def parse_columns(path):
    """{kernel: {counter: value}} of a multi-counter summary (tools/pmc_summary.py: header row names the counters, truncated to 16 characters)"""
    lines = open(path).read().splitlines()
    header = lines[0].split()
    counters = header[3:]
    out = {}
    for line in lines[1:]:
        fields = line.rsplit(None, len(counters) + 2)
        if len(fields) == len(counters) + 3:
            try:
                out[fields[0].strip()] = dict(zip(counters, [float(v) if v != "-" else None for v in fields[3:]]))
            except ValueError:
                pass
    return out


def main():
    key, fetch_file, write_file, source = sys.argv[1:5]
    sq = parse_columns(sys.argv[5]) if len(sys.argv) > 5 else {}
    sq_avg_us = {}
    if len(sys.argv) > 5:
        for line in open(sys.argv[5]).read().splitlines()[1:]:
            m = re.match(r"(.*?)\s+(\d+)\s+([\d.]+)\s+[\d.e+-]+", line)
            if m:
                sq_avg_us[m.group(1).strip()] = float(m.group(3))
    base = key[: -len("_nosky")] if key.endswith("_nosky") else key
    family = {"REBLUR_DIFFUSE_SPECULAR": "REBLUR_DiffuseSpecular_", "RELAX_DIFFUSE_SPECULAR_SH": "RELAX_DiffuseSpecularSh_", "RELAX_DIFFUSE_SPECULAR": "RELAX_DiffuseSpecular_"}[base.rsplit("_", 1)[0]]
    fetch, write = parse(fetch_file), parse(write_file)
    kernels = {}
    for kname, f in fetch.items():
        for frag, suffix in KERNEL_TO_SHADER:
            if frag in kname:
                shader = (family.split("_")[0] + "_" + suffix) if suffix == "ClassifyTiles.cs" else family + suffix
                w = next((v for k, v in write.items() if k == kname), None)
                if w is not None and shader not in kernels:  # (several variants of one pass, e.g. the a-trous steps: the first = longest-running one)
                    kernels[shader] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w}
                    c = sq.get(kname)
                    if c and c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES"):
                        kernels[shader].update({"SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_WAVES": c["SQ_WAVES"], "valu_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)})
                        # (round 3 also derived "cycles per instruction" and a "VALU busy" fraction from SQ_ACTIVE_INST_VALU; the counter advances in 4-cycle quanta, the
                        #  fractions read above 1.0 and are gone: profiles/issue_floor.json holds the measured floors instead -- DESIGN.md section 3.1)
                        if sq_avg_us.get(kname):
                            kernels[shader]["pmc_run_avg_us"] = sq_avg_us[kname]
                break
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = {"source": source, "library_digest": library_digest(), "kernels": kernels}
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(data[key], indent=1))


if __name__ == "__main__":
    main()
