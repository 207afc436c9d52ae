"""CPU: the GRC block descriptions (gr-clenabled_amd/grc/*.block.yml) against the C++ block API and the Python mirror.

What a saved flowgraph depends on is checked here: block ids, parameter ids (the reference's, SURVEY.md 2.1 row 12 -- frozen
below), that every ${...} of a template names a declared parameter, and that each `make` template calls the class with a
positional argument count the declared `make(...)` of host/include/clenabled/<Block>.h (and blocks.py) accepts."""
import glob
import inspect
import os
import re

import pytest
import yaml

from conftest import ROOT

GRC = os.path.join(ROOT, "gr-clenabled_amd", "grc")
HEADERS = os.path.join(ROOT, "gr-clenabled_amd", "host", "include", "clenabled", "cl*.h")  # one public header per block

DEVP = ["openCLPlatform", "devices", "platformId", "deviceId"]
DESIGN = DEVP + ["use_time", "decimation", "gain", "samp_rate"]
# parameter ids a flowgraph written for the reference carries (grc/clenabled_<X>.block.yml of the reference); extra ids here
# must be additive (with defaults)
REFERENCE_IDS = {
    "clFFT": ["type"] + DEVP + ["fft_dir", "fft_size", "window", "shift", "num_streams", "setDebug"],
    "clFIRTapFilter": DEVP + ["taps", "use_time", "decimation", "samp_rate", "setDebug"],
    "clLowPassFilter": DESIGN + ["cutoff_freq", "width", "win", "beta", "setDebug"],
    "clHighPassFilter": DESIGN + ["cutoff_freq", "width", "win", "beta", "setDebug"],
    "clBandPassFilter": DESIGN + ["low_cutoff_freq", "high_cutoff_freq", "width", "win", "beta", "setDebug"],
    "clBandRejectFilter": DESIGN + ["low_cutoff_freq", "high_cutoff_freq", "width", "win", "beta", "setDebug"],
    "clRootRaisedCosine": DEVP + ["use_time", "samp_rate", "sym_rate", "alpha", "ntaps", "setDebug"],
    "clComplexFilter": DEVP + ["taps", "decimation", "samp_rate", "setDebug"],
    "clAdd": ["type"] + DEVP + ["setDebug"], "clSubtract": ["type"] + DEVP + ["setDebug"], "clMultiply": ["type"] + DEVP + ["setDebug"],
    "clMultiplyConjugate": DEVP + ["setDebug"], "clComplexConjugate": DEVP + ["setDebug"],
    "clAddConst": ["type"] + DEVP + ["const", "setDebug"], "clMultConst": ["type"] + DEVP + ["const", "setDebug"],
    "clPolyphaseChannelizer": DEVP + ["taps", "buf_items", "num_channels", "ninputs_per_iter", "chmap", "setDebug"],
    "clXEngine": DEVP + ["type", "sync_timestamp", "first_channel", "starting_chan_center_freq", "num_channels", "channel_width", "num_inputs",
                         "polarization", "integration", "pipeline_integration", "output_file", "file_base", "rollover_size_mb",
                         "internal_synchronizer", "object_name", "antenna_list", "disable_output", "setDebug"],
    # widened rows
    "clLog10": DEVP + ["setDebug", "n_val", "k_val"], "clSNR": DEVP + ["setDebug", "n_val", "k_val"],
    "clComplexToMag": DEVP + ["setDebug"], "clComplexToArg": DEVP + ["setDebug"], "clComplexToMagPhase": DEVP + ["setDebug"],
    "clMagPhaseToComplex": DEVP + ["setDebug"], "clQuadratureDemod": DEVP + ["setDebug", "gain"],
    "clxcorrelate_fft_vcf": ["input_type", "vec_len", "num_inputs"] + DEVP,
}
# the block id of every description, frozen from the reference's grc/clenabled_<X>.block.yml line 3 (a saved flowgraph carries
# the id, not the file name; several of the reference's ids are spelled differently from their file stems)
BLOCK_ID = {
    "clFFT": "clenabled_clFFT", "clFIRTapFilter": "clenabled_cltapfirfilter", "clLowPassFilter": "clenabled_clLowPassFilter",
    "clHighPassFilter": "clenabled_clHighPassFilter", "clBandPassFilter": "clenabled_clBandPassFilter",
    "clBandRejectFilter": "clenabled_clBandRejectFilter", "clRootRaisedCosine": "clenabled_clRootRaisedCosineFilter",
    "clComplexFilter": "clenabled_clcomplexfilter", "clAdd": "clenabled_clAdd", "clSubtract": "clenabled_clSubtract",
    "clMultiply": "clenabled_clMultiply", "clMultiplyConjugate": "clenabled_clMultiplyConjugate",
    "clComplexConjugate": "clenabled_clComplexConjugate", "clAddConst": "clenabled_clAddConst", "clMultConst": "clenabled_clMultConst",
    "clPolyphaseChannelizer": "clenabled_clPolyphaseChannelizer", "clXEngine": "clenabled_clXEngine",
    "clLog10": "clenabled_clLog10", "clSNR": "clenabled_clsnr", "clComplexToMag": "clenabled_complextomag",
    "clComplexToArg": "clenabled_complextoarg", "clComplexToMagPhase": "clenabled_complextomagphase",
    "clMagPhaseToComplex": "clenabled_magphasetocomplex", "clQuadratureDemod": "clenabled_clQuadratureDemod",
    "clxcorrelate_fft_vcf": "clenabled_clxcorrelate_fft_vcf",
}
REFERENCE_GRC = "/root/reference/grc"  # present in the build container only; never read by the GPU tests


def split_args(s):
    """Top-level comma split of a call's argument text."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def header_signatures():
    """class name -> (required, total) positional parameters of its static make(), from the per-block public headers."""
    sigs = {}
    for path in sorted(glob.glob(HEADERS)):
        txt = open(path).read()
        for m in re.finditer(r"class\s+(?:CLENABLED_API\s+)?(\w+)\s*:[^{]*\{(.*?)\n\};", txt, re.S):
            mk = re.search(r"static\s+sptr\s+make\((.*?)\);", m.group(2), re.S)
            if mk:
                params = split_args(mk.group(1).replace("\n", " "))
                sigs[m.group(1)] = (sum("=" not in p for p in params), len(params))
    return sigs


def calls_of(template):
    """[(class, [args])] for every `clenabled.X(...)` line of a make template (both Mako branches)."""
    res = []
    for m in re.finditer(r"clenabled\.(\w+)\(", template):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(template[i], 0)
            i += 1
        res.append((m.group(1), split_args(template[m.end():i - 1].replace("\n", " ").replace("\t", " "))))
    return res


FILES = sorted(glob.glob(os.path.join(GRC, "clenabled_*.block.yml")))


def test_every_hot_path_block_has_a_description():
    names = {os.path.basename(f)[len("clenabled_"):-len(".block.yml")] for f in FILES}
    assert names == set(REFERENCE_IDS) == set(BLOCK_ID) and len(BLOCK_ID) == 25


@pytest.mark.skipif(not os.path.isdir(REFERENCE_GRC), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", sorted(BLOCK_ID))
def test_frozen_tables_equal_the_reference_files(name):
    """The frozen id / parameter tables above are what the reference's own block descriptions say (same file names)."""
    ref = yaml.safe_load(open(os.path.join(REFERENCE_GRC, "clenabled_%s.block.yml" % name)))
    assert ref["id"] == BLOCK_ID[name]
    ref_params = [p["id"] for p in ref.get("parameters", [])]
    assert sorted(ref_params) == sorted(REFERENCE_IDS[name]), (ref_params, REFERENCE_IDS[name])
    ours = yaml.safe_load(open(os.path.join(GRC, "clenabled_%s.block.yml" % name)))
    assert ours["id"] == ref["id"]
    # port layout a saved connection depends on: number of declared ports per domain, their dtypes and multiplicities
    for side in ("inputs", "outputs"):
        r, o = ref.get(side) or [], ours.get(side) or []
        assert [(x.get("domain"), str(x.get("dtype", "")).replace(" ", "")) for x in r] == \
               [(x.get("domain"), str(x.get("dtype", "")).replace(" ", "")) for x in o], side


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_block_yaml_contract(path, pkg):
    name = os.path.basename(path)[len("clenabled_"):-len(".block.yml")]
    d = yaml.safe_load(open(path))
    assert d["id"] == BLOCK_ID[name] and d["file_format"] == 1
    ids = [p["id"] for p in d["parameters"]]
    assert len(ids) == len(set(ids))
    missing = [i for i in REFERENCE_IDS[name] if i not in ids]
    assert not missing, "flowgraphs of the reference set %s" % missing
    for p in d["parameters"]:
        if p["id"] not in REFERENCE_IDS[name]:
            assert "default" in p, "an added parameter (%s) needs a default so old flowgraphs still load" % p["id"]
        if p.get("dtype") == "enum":
            assert p.get("options"), p["id"]
            if "option_labels" in p:
                assert len(p["option_labels"]) == len(p["options"])
            for k, v in (p.get("option_attributes") or {}).items():
                assert len(v) == len(p["options"]), (p["id"], k)
    # every ${...} names a declared parameter
    blob = yaml.safe_dump({k: d[k] for k in ("inputs", "outputs", "templates", "parameters", "asserts") if k in d})
    for ref in re.findall(r"\$\{\s*\(?\s*'?\w*'?\s*(?:if\s+)?(\w+)", blob):
        pass  # (expression forms are covered by the identifier scan below)
    for expr in re.findall(r"\$\{(.*?)\}", blob, re.S):
        for ident in re.findall(r"[A-Za-z_]\w*(?:\.\w+)?", re.sub(r"'[^']*'", "", expr)):
            base = ident.split(".")[0]
            if base in ("if", "else", "or", "and", "not", "True", "False", "replace", "split", "all", "none", "part"):
                continue
            assert base in ids, "%s: ${%s} refers to an undeclared parameter" % (name, expr.strip())
    # make templates: class exists in the C++ API, positional count fits its make()
    sigs = header_signatures()
    calls = calls_of(d["templates"]["make"])
    assert len(calls) == 2, "one call per Mako branch"
    assert calls[0][0] == calls[1][0] and len(calls[0][1]) == len(calls[1][1])
    for cls, args in calls:
        assert cls in sigs, "%s is not declared in clenabled.h" % cls
        lo, hi = sigs[cls]
        assert lo <= len(args) <= hi, "%s: %d positional arguments, make() takes %d..%d" % (cls, len(args), lo, hi)
        # the Python mirror (what `import clenabled` resolves to in this repo's tests) accepts the same call shape
        ctor = inspect.signature(getattr(pkg, cls).__init__)
        pos = [p for p in list(ctor.parameters.values())[1:] if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        need = sum(p.default is p.empty for p in pos)
        assert need <= len(args) <= len(pos), "%s (python mirror): %d arguments, takes %d..%d" % (cls, len(args), need, len(pos))
    # the device-selection slots: "first device" branch passes selector 1 and ids 0,0
    first, chosen = calls[0][1], calls[1][1]
    k = next(i for i, a in enumerate(first) if "openCLPlatform" in a)
    assert first[k + 1:k + 4] == ["1", "0", "0"]
    assert [re.sub(r"\s", "", a) for a in chosen[k + 1:k + 4]] == ["${devices}", "${platformId}", "${deviceId}"]
    for cb in d["templates"].get("callbacks") or []:
        assert re.match(r"set_(taps2|k)\(", cb)


def test_generator_output_is_current():
    """The committed files are what grc/gen_block_yml.py writes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_block_yml", os.path.join(GRC, "gen_block_yml.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for b in mod.BLOCKS:
        with open(os.path.join(GRC, b.get("file", b["id"]) + ".block.yml")) as f:
            assert f.read() == mod.render(b), b["id"]
