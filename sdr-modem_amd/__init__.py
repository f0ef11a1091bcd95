"""sdr-modem_amd: MI355X-native GMSK/FSK demodulation path (HIP kernels + C-ABI + host mirror of the
reference's operator / worker surface).

The directory name is not a valid Python identifier; import it through the repo-root helper:

    import sdrm_pkg; sdrm = sdrm_pkg.load()       # registers the package as `sdr_modem_amd`
    from sdr_modem_amd import binding, siggen

The compute lives in csrc/libsdrmodem_hip.so (built by csrc/Makefile / __graft_entry__.build()).  This package
never falls back to a CPU implementation: binding.load() raises if the library is missing.
"""
