# Recipe for the one build this repository's development image cannot do: the out-of-tree module against a REAL GNU Radio.
# (SURVEY 8f-1: "built against a real GNU Radio 3.9 / 3.10".  The development image has ROCm but no GNU Radio and no network, so the
# MI355_WITH_GNURADIO branches are only type-checked there -- tests/test_host_cpp.py::test_gnuradio_branch_compiles_against_the_api_model
# against tests/gr_api_mock/.  Every branch this recipe compiles is one that test already covers.)
#
#   docker build -f gr-clenabled_amd/ci/gnuradio.Dockerfile -t gr-clenabled-mi355 .          (context: the repository root)
#   docker run --rm --device=/dev/kfd --device=/dev/dri --group-add video gr-clenabled-mi355 gr-clenabled_amd/ci/check_flowgraphs.sh
#
# The build stage needs no GPU (hipcc cross-compiles gfx950); the run stage needs an MI355X.
ARG ROCM_IMAGE=rocm/dev-ubuntu-22.04:7.2
FROM ${ROCM_IMAGE}

# GNU Radio 3.10 from the distribution (Ubuntu 22.04 ships 3.10.1) with its development files, pybind11 and the tools CMake asks for
RUN apt-get update && DEBIAN_FRONTEND=noninteractive apt-get install -y --no-install-recommends \
        gnuradio gnuradio-dev libspdlog-dev libfmt-dev libboost-all-dev libvolk2-dev libfftw3-dev \
        pybind11-dev python3-dev python3-numpy python3-pytest python3-yaml cmake ninja-build g++ git \
    && rm -rf /var/lib/apt/lists/*

WORKDIR /src
COPY . /src

# 1. the module: find_package(Gnuradio) succeeds here, so CMake defines MI355_WITH_GNURADIO, the blocks derive from gr::sync_block /
#    gr::sync_decimator / gr::block, the pybind11 module registers them as GNU Radio blocks and the GRC descriptions are installed where
#    GNU Radio Companion looks for them
RUN cmake -S gr-clenabled_amd -B /build -G Ninja -DCMAKE_PREFIX_PATH=/opt/rocm -DCMAKE_INSTALL_PREFIX=/usr -DMI355_GPU_ARCH=gfx950 \
    && cmake --build /build \
    && cmake --install /build \
    && ldconfig

# 2. what can be checked without a GPU: the module imports next to GNU Radio's own, every block class is a gr block, and grcc generates
#    Python for one flowgraph per block description (block ids / parameter ids / make templates are the reference's)
RUN python3 -c "from gnuradio import gr; import clenabled; print('gnuradio', gr.version(), '- clenabled blocks:', sorted(n for n in dir(clenabled) if n.startswith('cl')))"
RUN python3 gr-clenabled_amd/ci/make_flowgraphs.py /tmp/flowgraphs && for f in /tmp/flowgraphs/*.grc; do grcc -o /tmp/flowgraphs "$f" || exit 1; done

# 3. (run stage, on an MI355X) the generated flowgraphs run for a second each and the repository's GPU suite passes inside this image
CMD ["gr-clenabled_amd/ci/check_flowgraphs.sh"]
