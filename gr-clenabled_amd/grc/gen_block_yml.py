#!/usr/bin/env python3
"""Writes grc/clenabled_<X>.block.yml for the hot-path blocks and the widened rows (8f-3 / 8f-4).

What an existing flowgraph stores is the block id, the parameter ids with their values, and the port layout; what it needs
from the block description is a `make` template that turns those into `clenabled.<class>(...)` with the reference's
positional order.  Those are kept (ids, keys, option values, port multiplicities, positional order of every make call);
labels, help text and layout are this build's.  Deliberate differences, all additive or corrective:
  * device ids 0-7 (an MI355X node has eight GPUs; the reference offers 0-3);
  * clAddConst / clMultConst pass ${const} in the "any device" branch too (the reference drops it there, so the operator
    code lands in the constant's slot: SURVEY App. B-5);
  * clRootRaisedCosine declares the `gain` and `decimation` parameters its own make template reads (undefined in the
    reference's file);
  * the set_taps2 callbacks call gnuradio's firdes, not `clenabled.firdes` (which does not exist in the Python module).
Run:  python gr-clenabled_amd/grc/gen_block_yml.py   (tests/test_grc_yaml.py checks the result against clenabled.h).
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CATEGORY = "[MI355X clenabled]"

DEV = [  # the four device-selection parameters every block starts with (positional slots 2-5 of every make())
    dict(id="openCLPlatform", label="Device class", dtype="enum", options=["1", "2", "3", "4"],
         option_labels=["GPU (MI355X)", "Accelerator (same)", "CPU (refused)", "Any"]),
    dict(id="devices", label="Device choice", dtype="enum", options=["1", "2"], option_labels=["First", "By id"],
         option_attributes={"hide_specific": ["all", ""]}),
    dict(id="platformId", label="Platform id (ignored)", dtype="enum", options=["0", "1", "2", "3"], hide="${ devices.hide_specific }"),
    dict(id="deviceId", label="GPU ordinal", dtype="enum", options=[str(i) for i in range(8)], hide="${ devices.hide_specific }"),
]
DEBUG = dict(id="setDebug", label="Verbose", dtype="enum", options=["0", "1"], option_labels=["No", "Yes"])
TYPE3 = dict(id="type", label="Item type", dtype="enum", options=["complex", "float", "int"], hide="part",
             option_attributes={"datatype": ["1", "2", "3"], "input": ["complex", "float", "int"], "output": ["complex", "float", "int"]})
MODE = dict(id="use_time", label="Kernel", dtype="enum", options=["True", "False"],
            option_labels=["Direct form (time domain)", "Overlap-save (frequency domain)"])


def two_branch(call_any, call_id):
    """The reference's Mako shape: `devices == 1` -> first device (selector 1, ids 0,0), else the chosen ids."""
    return "% if devices == 1:\n" + call_any + "\n% else:\n" + call_id + "\n% endif"


def dev_args(any_branch):
    return "${openCLPlatform},1,0,0" if any_branch else "${openCLPlatform},${devices},${platformId},${deviceId}"


def mathop(block, label, op, typed=True, doc=""):
    dt = "${type.datatype}" if typed else "1"
    ports_t = "${ type.input }" if typed else "complex"
    make = two_branch("clenabled.clMathOp(%s,%s,%d,${setDebug})" % (dt, dev_args(True), op),
                      "clenabled.clMathOp(%s,%s,%d,${setDebug})" % (dt, dev_args(False), op))
    return dict(id="clenabled_" + block, label=label, params=([TYPE3] if typed else []) + DEV + [DEBUG],
                inputs=[dict(label="in1", domain="stream", dtype=ports_t), dict(label="in2", domain="stream", dtype=ports_t)],
                outputs=[dict(domain="stream", dtype="${ type.output }" if typed else "complex")], imports="import clenabled", make=make, doc=doc)


def mathconst(block, label, op, typed=True, doc=""):
    dt = "${type.datatype}" if typed else "1"
    k = "${const}" if typed else "0"
    make = two_branch("clenabled.clMathConst(%s,%s,%s,%d,${setDebug})" % (dt, dev_args(True), k, op),
                      "clenabled.clMathConst(%s,%s,%s,%d,${setDebug})" % (dt, dev_args(False), k, op))
    params = ([TYPE3] if typed else []) + DEV + ([dict(id="const", label="Constant", dtype="${ type.input }", default="1")] if typed else []) + [DEBUG]
    t_in = "${ type.input }" if typed else "complex"
    return dict(id="clenabled_" + block, label=label, params=params, inputs=[dict(domain="stream", dtype=t_in)],
                outputs=[dict(domain="stream", dtype="${ type.output }" if typed else "complex")], imports="import clenabled", make=make,
                callbacks=["set_k(${const})"] if typed else None, doc=doc)


def design_filter(block, label, design, extra_params, doc, win_raw=False, bid=None):
    """The five filter-design front-ends: all construct clenabled.clFilter with gnuradio firdes taps."""
    call = "firdes.%s(%s)" % (design[0], ", ".join("${%s}" % a for a in design[1]))
    make = two_branch("clenabled.clFilter(%s,${decimation},%s,1,${setDebug},${use_time})" % (dev_args(True), call),
                      "clenabled.clFilter(%s,${decimation},%s,1,${setDebug},${use_time})" % (dev_args(False), call))
    imports = "import clenabled\nfrom gnuradio.filter import firdes" + ("\nfrom gnuradio.filter import window" if win_raw else "")
    return dict(id=bid or "clenabled_" + block, file="clenabled_" + block, label=label, params=DEV + [MODE] + extra_params + [DEBUG],
                inputs=[dict(domain="stream", dtype="complex")], outputs=[dict(domain="stream", dtype="complex")], imports=imports, make=make,
                callbacks=["set_taps2(%s)" % call], doc=doc)


def win_param(raw):
    if raw:  # the low-pass block of the reference takes gnuradio.filter.window constants
        opts = ["window.WIN_HAMMING", "window.WIN_HANN", "window.WIN_BLACKMAN", "window.WIN_RECTANGULAR", "window.WIN_KAISER"]
        return dict(id="win", label="Window", dtype="raw", default=opts[0], options=opts,
                    option_labels=["Hamming", "Hann", "Blackman", "Rectangular", "Kaiser"])
    opts = ["firdes.WIN_HAMMING", "firdes.WIN_BLACKMAN", "firdes.WIN_HANN", "firdes.WIN_RECTANGULAR", "firdes.WIN_KAISER"]
    return dict(id="win", label="Window", dtype="int", default=opts[0], options=opts,
                option_labels=["Hamming", "Blackman", "Hann", "Rectangular", "Kaiser"])


REAL = lambda i, l, d=None: dict(id=i, label=l, dtype="real", **({"default": d} if d is not None else {}))
INT = lambda i, l, d=None: dict(id=i, label=l, dtype="int", **({"default": d} if d is not None else {}))
COMMON_DESIGN = [INT("decimation", "Decimation", "1"), REAL("gain", "Gain", "1"), REAL("samp_rate", "Sample rate", "samp_rate")]

BLOCKS = []
BLOCKS.append(dict(
    id="clenabled_clFFT", label="MI355X FFT",
    params=[dict(id="type", label="Input type", dtype="enum", options=["complex", "float"], hide="part",
                 option_attributes={"datatype": ["1", "2"], "input": ["complex", "float"], "output": ["complex", "complex"]})] + DEV + [
        dict(id="fft_dir", label="Direction", dtype="enum", options=["-1", "1"], option_labels=["Forward", "Reverse"]),
        INT("fft_size", "Points", "1024"),
        dict(id="window", label="Window taps", dtype="real_vector", default="window.blackmanharris(1024)"),
        dict(id="shift", label="fftshift", dtype="enum", options=["True", "False"], option_labels=["On", "Off"],
             hide="${ 'all' if type == 'float' else 'none' }"),
        dict(id="num_streams", label="Parallel streams", dtype="int", default="1", hide="part"),
        dict(DEBUG, hide="part")],
    inputs=[dict(domain="stream", dtype="${ type.input }", vlen="${ fft_size }", multiplicity="${num_streams}")],
    outputs=[dict(domain="stream", dtype="complex", vlen="${ fft_size }", multiplicity="${num_streams}")],
    imports="from gnuradio.fft import window\nimport clenabled",
    make=two_branch("clenabled.clFFT(${fft_size},${fft_dir},${window},${type.datatype},%s,${setDebug},${num_streams},${shift})" % dev_args(True),
                    "clenabled.clFFT(${fft_size},${fft_dir},${window},${type.datatype},%s,${setDebug},${num_streams},${shift})" % dev_args(False)),
    doc="Vector FFT on the GPU: window multiply, transform and fftshift in one kernel (any power of two up to 16777216; other "
        "lengths up to 8388608 by chirp-z).  One item = one vector of `Points` samples."))
BLOCKS.append(dict(
    id="clenabled_cltapfirfilter", file="clenabled_clFIRTapFilter", label="MI355X FIR filter (given taps)",
    params=DEV + [dict(id="taps", label="Taps", dtype="real_vector"), MODE, INT("decimation", "Decimation", "1"),
                  REAL("samp_rate", "Sample rate", "samp_rate"), DEBUG],
    inputs=[dict(domain="stream", dtype="complex")], outputs=[dict(domain="stream", dtype="complex")],
    imports="import clenabled\nfrom gnuradio.filter import firdes",
    make=two_branch("clenabled.clFilter(%s,${decimation},${taps},1,${setDebug},${use_time})" % dev_args(True),
                    "clenabled.clFilter(%s,${decimation},${taps},1,${setDebug},${use_time})" % dev_args(False)),
    callbacks=["set_taps2(${taps})"],
    doc="Complex stream, real taps.  Output equals gnuradio's fir_filter_ccf / fft_filter_ccf on the same taps."))
BLOCKS.append(design_filter("clLowPassFilter", "MI355X low-pass filter", ("low_pass", ["gain", "samp_rate", "cutoff_freq", "width", "win", "beta"]),
                            COMMON_DESIGN + [REAL("cutoff_freq", "Cutoff"), REAL("width", "Transition width"), win_param(True), REAL("beta", "Kaiser beta", "6.76")],
                            "firdes.low_pass taps into the GPU filter.", win_raw=True))
BLOCKS.append(design_filter("clHighPassFilter", "MI355X high-pass filter", ("high_pass", ["gain", "samp_rate", "cutoff_freq", "width", "win", "beta"]),
                            COMMON_DESIGN + [REAL("cutoff_freq", "Cutoff"), REAL("width", "Transition width"), win_param(False), REAL("beta", "Kaiser beta", "6.76")],
                            "firdes.high_pass taps into the GPU filter."))
BLOCKS.append(design_filter("clBandPassFilter", "MI355X band-pass filter",
                            ("complex_band_pass", ["gain", "samp_rate", "low_cutoff_freq", "high_cutoff_freq", "width", "win", "beta"]),
                            COMMON_DESIGN + [REAL("low_cutoff_freq", "Lower edge"), REAL("high_cutoff_freq", "Upper edge"), REAL("width", "Transition width"),
                                             win_param(False), REAL("beta", "Kaiser beta", "6.76")],
                            "firdes.complex_band_pass taps into the GPU filter."))
BLOCKS.append(design_filter("clBandRejectFilter", "MI355X band-reject filter",
                            ("band_reject", ["gain", "samp_rate", "low_cutoff_freq", "high_cutoff_freq", "width", "win", "beta"]),
                            COMMON_DESIGN + [REAL("low_cutoff_freq", "Lower edge"), REAL("high_cutoff_freq", "Upper edge"), REAL("width", "Transition width"),
                                             win_param(False), REAL("beta", "Kaiser beta", "6.76")],
                            "firdes.band_reject taps into the GPU filter."))
BLOCKS.append(design_filter("clRootRaisedCosine", "MI355X root-raised-cosine filter",
                            ("root_raised_cosine", ["gain", "samp_rate", "sym_rate", "alpha", "ntaps"]),
                            COMMON_DESIGN + [REAL("sym_rate", "Symbol rate", "1.0"), REAL("alpha", "Roll-off", "0.35"), INT("ntaps", "Taps", "11*samp_rate")],
                            "firdes.root_raised_cosine taps into the GPU filter.", bid="clenabled_clRootRaisedCosineFilter"))
BLOCKS.append(dict(
    id="clenabled_clcomplexfilter", file="clenabled_clComplexFilter", label="MI355X FIR filter (complex taps)",
    params=DEV + [dict(id="taps", label="Taps", dtype="complex_vector"), INT("decimation", "Decimation", "1"),
                  REAL("samp_rate", "Sample rate", "samp_rate"), DEBUG],
    inputs=[dict(domain="stream", dtype="complex")], outputs=[dict(domain="stream", dtype="complex")],
    imports="import clenabled\nfrom gnuradio.filter import firdes",
    make=two_branch("clenabled.clComplexFilter(%s,${decimation},${taps},1,${setDebug})" % dev_args(True),
                    "clenabled.clComplexFilter(%s,${decimation},${taps},1,${setDebug})" % dev_args(False)),
    callbacks=["set_taps2(${taps})"], doc="Complex stream, complex taps (gnuradio's fir_filter_ccc)."))
BLOCKS.append(mathop("clAdd", "MI355X add", 2, doc="out = in1 + in2"))
BLOCKS.append(mathop("clSubtract", "MI355X subtract", 3, doc="out = in1 - in2"))
BLOCKS.append(mathop("clMultiply", "MI355X multiply", 1, doc="out = in1 * in2"))
BLOCKS.append(mathop("clMultiplyConjugate", "MI355X multiply by conjugate", 5, typed=False, doc="out = in1 * conj(in2)"))
BLOCKS.append(mathconst("clComplexConjugate", "MI355X complex conjugate", 4, typed=False, doc="out = conj(in)"))
BLOCKS.append(mathconst("clAddConst", "MI355X add constant", 2, doc="out = in + k (a complex item gets k on both components, as in the reference)"))
BLOCKS.append(mathconst("clMultConst", "MI355X multiply by constant", 1, doc="out = k * in"))
BLOCKS.append(dict(
    id="clenabled_clPolyphaseChannelizer", label="MI355X polyphase channelizer",
    params=DEV + [dict(id="taps", label="Prototype taps", dtype="real_vector"), INT("buf_items", "Items per call"), INT("num_channels", "Channels"),
                  INT("ninputs_per_iter", "Inputs per output step"), dict(id="chmap", label="Output channel map", dtype="int_vector"), DEBUG],
    inputs=[dict(domain="stream", dtype="complex")], outputs=[dict(domain="stream", dtype="complex")], imports="import clenabled",
    make=two_branch("clenabled.clPolyphaseChannelizer(%s, ${taps}, ${buf_items}, ${num_channels}, ${ninputs_per_iter}, ${chmap})" % dev_args(True).replace(",", ", "),
                    "clenabled.clPolyphaseChannelizer(%s, ${taps}, ${buf_items}, ${num_channels}, ${ninputs_per_iter}, ${chmap})" % dev_args(False).replace(",", ", ")),
    doc="Critically sampled or oversampled polyphase filter bank; channel 0 is the centre frequency, higher indices are "
        "higher frequencies (wrapping).  `Items per call` must be a multiple of the channel count."))
XE_TAIL = ("${setDebug}, ${type.data_type}, ${polarization}, ${num_inputs}, 1, ${first_channel}, ${num_channels}, ${integration}, "
           "${antenna_list}.replace(' ','').split(','), ${output_file},${file_base},${rollover_size_mb},${internal_synchronizer}, "
           "${sync_timestamp}, ${object_name}, ${starting_chan_center_freq}, ${channel_width}, ${disable_output}, ${pipeline_integration})")
ONOFF = lambda i, l, hide=None: dict(id=i, label=l, dtype="enum", options=["False", "True"], option_labels=["Off", "On"], **({"hide": hide} if hide else {}))
VLEN = "${ (num_channels if type.data_type == 1 else num_channels*2) }"
NO_POL2 = "((polarization == '1') or (type.data_type == 6))"
BLOCKS.append(dict(
    id="clenabled_clXEngine", label="MI355X X-engine (FX correlator)",
    params=[dict(DEV[0], hide="part"), dict(DEV[1], hide="part", option_attributes={"hide_specific": ["all", "part"]}), DEV[2], DEV[3],
            dict(id="type", label="Sample format", dtype="enum", options=["Complex", "IChar", "Packed XY"], hide="part",
                 option_attributes={"data_type": [1, 5, 6], "input_format": ["complex", "char", "char"]}),
            dict(id="sync_timestamp", label="Start timestamp", dtype="int", default="0", hide="part"),
            INT("first_channel", "First channel", "0"),
            dict(id="starting_chan_center_freq", label="First channel centre (Hz)", dtype="float", default="0", hide="part"),
            INT("num_channels", "Channels", "256"),
            dict(id="channel_width", label="Channel width (Hz)", dtype="float", default="0", hide="part"),
            INT("num_inputs", "Antennas", "2"),
            dict(id="polarization", label="Polarisations", dtype="enum", options=["1", "2"], option_labels=["One", "X and Y"]),
            INT("integration", "Frames per integration", "10000"),
            dict(id="pipeline_integration", label="Integrations summed on the host", dtype="int", default="0", hide="part"),
            dict(id="output_file", label="Result goes to", dtype="enum", options=["False", "True"], option_labels=["Message port", "File"]),
            dict(id="file_base", label="File name stem", dtype="string", hide="${ 'none' if output_file=='True' else 'all' }"),
            dict(id="rollover_size_mb", label="New file every (MB)", dtype="int", default="0", hide="${ 'part' if output_file=='True' else 'all' }"),
            ONOFF("internal_synchronizer", "Tag synchroniser", "part"),
            dict(id="object_name", label="Object", dtype="string", default="", hide="part"),
            dict(id="antenna_list", label="Antenna names", dtype="string", default="", hide="part"),
            ONOFF("disable_output", "Discard results", "part"), ONOFF("setDebug", "Verbose", "part")],
    inputs=[dict(label="pol1_", domain="stream", dtype="${ type.input_format }", vlen=VLEN, multiplicity="${num_inputs}"),
            dict(label="pol2_", domain="stream", dtype="${ type.input_format }", vlen=VLEN,
                 multiplicity="${ (0 if %s else num_inputs) }" % NO_POL2, optional="${ (True if %s else False) }" % NO_POL2,
                 hide="${ (True if %s else False) }" % NO_POL2)],
    outputs=[dict(label="xcorr", domain="message", optional=True, hide="${ output_file }"), dict(label="sync", domain="message", optional=True)],
    imports="import clenabled", asserts=["${ num_inputs > 1 }"],
    make=two_branch("clenabled.clXEngine(%s," % dev_args(True) + XE_TAIL, "clenabled.clXEngine(%s, " % dev_args(False) + XE_TAIL),
    doc="Cross-correlates every antenna pair per channel over `Frames per integration` frames (int8 and packed 4-bit inputs with exact "
        "integer sums on the matrix cores).  Output: lower-triangular baseline order per channel, as a message or appended to a file "
        "with a JSON side-car."))


# ---- widened rows: the remaining elementwise blocks and the reference correlator.  The ids are the
# reference's (lower case where its files have them so: a saved flowgraph names the id, not the file).
def elem(file, bid, label, cls, ins, outs, extra=(), lead="", doc=""):
    tail = "".join(",${%s}" % e["id"] for e in extra)
    make = two_branch("clenabled.%s(%s%s%s,${setDebug})" % (cls, lead, dev_args(True), tail),
                      "clenabled.%s(%s%s%s,${setDebug})" % (cls, lead, dev_args(False), tail))
    return dict(id=bid, file=file, label=label, params=DEV + [DEBUG] + [e for e in extra] + ([GAIN] if lead else []),
                inputs=[dict(domain="stream", dtype=t, **({"label": l} if l else {})) for l, t in ins],
                outputs=[dict(domain="stream", dtype=t, **({"label": l} if l else {})) for l, t in outs], imports="import clenabled", make=make, doc=doc)


GAIN = dict(id="gain", label="Gain", dtype="float", default="1.0")
NK = [dict(id="n_val", label="n", dtype="float", default="1"), dict(id="k_val", label="k", dtype="float", default="0")]
BLOCKS.append(elem("clenabled_clLog10", "clenabled_clLog10", "MI355X n*log10(x)+k", "clLog", [("", "float")], [("", "float")], NK,
                   doc="out = n * log10(in) + k"))
BLOCKS.append(elem("clenabled_clSNR", "clenabled_clsnr", "MI355X SNR helper", "clSNR", [("signal", "float"), ("noise", "float")], [("", "float")], NK,
                   doc="out = abs(n * log10(signal / noise) + k)"))
BLOCKS.append(elem("clenabled_clComplexToMag", "clenabled_complextomag", "MI355X complex to magnitude", "clComplexToMag", [("", "complex")], [("", "float")]))
BLOCKS.append(elem("clenabled_clComplexToArg", "clenabled_complextoarg", "MI355X complex to phase", "clComplexToArg", [("", "complex")], [("", "float")]))
BLOCKS.append(elem("clenabled_clComplexToMagPhase", "clenabled_complextomagphase", "MI355X complex to magnitude and phase", "clComplexToMagPhase",
                   [("", "complex")], [("mag", "float"), ("phase", "float")]))
BLOCKS.append(elem("clenabled_clMagPhaseToComplex", "clenabled_magphasetocomplex", "MI355X magnitude and phase to complex", "clMagPhaseToComplex",
                   [("mag", "float"), ("phase", "float")], [("", "complex")]))
BLOCKS.append(elem("clenabled_clQuadratureDemod", "clenabled_clQuadratureDemod", "MI355X quadrature demodulator", "clQuadratureDemod",
                   [("", "complex")], [("", "float")], lead="${gain},", doc="out[i] = gain * arg(in[i] * conj(in[i-1]))"))
BLOCKS.append(dict(
    id="clenabled_clxcorrelate_fft_vcf", label="MI355X reference correlator (frequency domain)",
    params=[dict(id="input_type", label="Inputs are", dtype="enum", options=["1", "2"], option_labels=["Spectra", "Time series"], hide="part"),
            dict(id="vec_len", label="Vector length", dtype="int", default="1024", hide="none"), INT("num_inputs", "Signals", "2")] + DEV,
    inputs=[dict(domain="stream", dtype="complex", vlen="${vec_len}", multiplicity="${ num_inputs }")],
    outputs=[dict(domain="stream", dtype="float", vlen="${vec_len}", multiplicity="${ num_inputs - 1 }")], imports="import clenabled",
    make=two_branch("clenabled.clxcorrelate_fft_vcf(${vec_len},${num_inputs},%s,${input_type})" % dev_args(True),
                    "clenabled.clxcorrelate_fft_vcf(${vec_len},${num_inputs},%s,${input_type})" % dev_args(False)),
    doc="Input 0 is the reference; output s-1 is the half-swapped magnitude of the inverse transform of X0 * conj(Xs)."))


def emit_value(v, indent):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return str(v)
    if isinstance(v, list):
        return "[" + ", ".join(emit_value(x, indent) for x in v) + "]"
    s = str(v)
    if "\n" in s:
        pad = " " * (indent + 2)
        return "|-\n" + "\n".join(pad + ln for ln in s.split("\n"))
    return "'" + s.replace("'", "''") + "'"


def emit_map_list(name, items, out):
    out.append("%s:" % name)
    for it in items:
        first = True
        for k, v in it.items():
            lead = "  - " if first else "    "
            first = False
            if isinstance(v, dict):
                out.append("%s%s:" % (lead, k))
                for kk, vv in v.items():
                    out.append("      %s: %s" % (kk, emit_value(vv, 6)))
            else:
                out.append("%s%s: %s" % (lead, k, emit_value(v, 4)))


def render(b):
    out = ["# generated by gr-clenabled_amd/grc/gen_block_yml.py -- edit the table there", "id: %s" % b["id"], "label: %s" % emit_value(b["label"], 0),
           "category: %s" % emit_value(CATEGORY, 0), ""]
    emit_map_list("parameters", b["params"], out)
    out.append("")
    emit_map_list("inputs", b["inputs"], out)
    out.append("")
    emit_map_list("outputs", b["outputs"], out)
    if b.get("asserts"):
        out += ["", "asserts:"] + ["  - %s" % emit_value(x, 2) for x in b["asserts"]]
    out += ["", "templates:", "  imports: %s" % emit_value(b["imports"], 2), "  make: %s" % emit_value(b["make"], 2)]
    if b.get("callbacks"):
        out += ["  callbacks:"] + ["    - %s" % emit_value(c, 4) for c in b["callbacks"]]
    out += ["", "documentation: %s" % emit_value(b.get("doc", ""), 0), "", "file_format: 1", ""]
    return "\n".join(out)


def main():
    for b in BLOCKS:
        with open(os.path.join(HERE, b.get("file", b["id"]) + ".block.yml"), "w") as f:
            f.write(render(b))
    print("%d block descriptions written to %s" % (len(BLOCKS), HERE))


if __name__ == "__main__":
    main()
