#!/usr/bin/env python3
"""One GNU Radio Companion flowgraph per hot-path block description (grc/clenabled_*.block.yml), written as .grc files for `grcc`.

Part of ci/gnuradio.Dockerfile: the development image of this repository has no GNU Radio, so these files are GENERATED inside the
image that has one.  Every flowgraph is  null sources -> [head] -> the block under test (its defaults, first device) -> null sinks,
`run` to completion without a GUI; what it proves is that the block id, the parameter ids and the make template of the description
produce Python that GNU Radio's own generator accepts and its scheduler runs.

usage: make_flowgraphs.py <output directory>
"""
import os
import sys

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
GRC = os.path.join(os.path.dirname(HERE), "grc")

# per block id: parameter overrides (everything else keeps the description's default / first option), and the stream types / vector
# lengths of its ports as (type, vlen) -- the same values the overrides imply
CASES = {
    "clenabled_clMultiply": (dict(type="complex"), [("complex", 1)] * 2, [("complex", 1)]),
    "clenabled_clMultConst": (dict(type="complex", const="2"), [("complex", 1)], [("complex", 1)]),
    "clenabled_clFFT": (dict(type="complex", fft_size="4096", window="window.blackman(4096)", shift="True", fft_dir="-1"),
                        [("complex", 4096)], [("complex", 4096)]),
    "clenabled_clFIRTapFilter": (dict(taps="firdes.low_pass(1.0, 10e6, 1e6, 372000.0)", decimation="1"), [("complex", 1)], [("complex", 1)]),
    "clenabled_clComplexFilter": (dict(taps="[complex(t) for t in firdes.low_pass(1.0, 10e6, 1e6, 372000.0)]", decimation="1"),
                                  [("complex", 1)], [("complex", 1)]),
    "clenabled_clPolyphaseChannelizer": (dict(taps="list(firdes.low_pass(1.0, 64.0, 0.5, 0.0753)) + [0.0]", num_channels="64",
                                              ninputs_per_iter="64", buf_items="65536", chmap="list(range(64))"),
                                         [("complex", 1)], [("complex", 1)]),
    "clenabled_clXEngine": (dict(type="IChar", polarization="1", num_inputs="4", num_channels="64", integration="32"),
                            [("byte", 128)] * 4, []),
}


def block(name, bid, params, x, y):
    return {"name": name, "id": bid, "parameters": {k: str(v) for k, v in params.items()},
            "states": {"bus_sink": False, "bus_source": False, "bus_structure": None, "coordinate": [x, y], "rotation": 0, "state": "enabled"}}


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    for bid, (over, ins, outs) in CASES.items():
        desc = yaml.safe_load(open(os.path.join(GRC, bid + ".block.yml")))
        params = {}
        for p in desc.get("parameters", []):
            params[p["id"]] = p.get("default", (p.get("options") or [""])[0])
        unknown = set(over) - set(params)
        assert not unknown, "%s: the description has no parameter(s) %s" % (bid, sorted(unknown))
        params.update(over)
        blocks, conns = [], []
        blocks.append(block("imp_firdes", "import", {"imports": "from gnuradio.filter import firdes; from gnuradio.fft import window"}, 8, 80))
        blocks.append(block("dut", bid, params, 400, 200))
        for i, (t, v) in enumerate(ins):
            blocks.append(block("src%d" % i, "blocks_null_source", dict(type=t, vlen=v, num_outputs=1, bus_structure_source="[[0,],]"), 8, 160 + 80 * i))
            blocks.append(block("head%d" % i, "blocks_head", dict(type=t, vlen=v, num_items=max(4, 4 * 65536 // v)), 200, 160 + 80 * i))
            conns += [["src%d" % i, "0", "head%d" % i, "0"], ["head%d" % i, "0", "dut", str(i)]]
        for i, (t, v) in enumerate(outs):
            blocks.append(block("snk%d" % i, "blocks_null_sink", dict(type=t, vlen=v, num_inputs=1, bus_structure_sink="[[0,],]"), 640, 160 + 80 * i))
            conns.append(["dut", str(i), "snk%d" % i, "0"])
        fg = {"options": {"parameters": {"id": "fg_" + bid, "title": bid, "generate_options": "no_gui", "run_options": "run", "output_language": "python",
                                         "category": "[GRC Hier Blocks]", "author": "", "description": "", "realtime_scheduling": "", "run": "True",
                                         "max_nouts": "0", "thread_safe_setters": "", "sizing_mode": "fixed", "catch_exceptions": "True"},
                          "states": {"bus_sink": False, "bus_source": False, "bus_structure": None, "coordinate": [8, 8], "rotation": 0, "state": "enabled"}},
              "blocks": blocks, "connections": conns, "metadata": {"file_format": 1, "grc_version": "3.10.1.1"}}
        with open(os.path.join(out_dir, "fg_%s.grc" % bid), "w") as f:
            yaml.safe_dump(fg, f, sort_keys=False)
        print("wrote", f.name)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "flowgraphs")
