"""`vgen-hip verify` (src/lib.rs:377-492): key as WIF or hex, every address of the key, --address matching.  Host code only
(vgen_derive): runs without a GPU.  Vectors: the reference README's key (README.md:121-127) and tests/golden."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "vgen_amd", "vgen-hip")
HEX = "0c28fca386c7a227600b2fe50b7cae11ec86d3bf1fbe471be89827e19d72aa1d"
WIF_U = "5HueCGU8rMjxEXxiPuD5BDku4MkFqeZyd4dZ1jvhTVqvbTLvyTJ"
WIF_C = "KwdMAjGmerYanjeui5SHS7JkmpZvVipYvB2LJGU1ZxJwYvP98617"
ADDRS = {
    "P2PKH address:": "1LoVGDgRs9hTfTNJNuXKSpywcbdvwRXpmK",
    "P2PKH (uncompr.):": "1GAehh7TsJAHuUAeKZcXf5CnwuGuGgyX2S",
    "P2WPKH address:": "bc1qmy63mjadtw8nhzl69ukdepwzsyvv4yex5qlmkd",
    "P2SH-P2WPKH addr:": "3D9iyFHi1Zs9KoyynUfrL82rGhJfYTfSG4",
    "P2TR address:": "bc1pdj78vjhv4wukfzfu3qyvwclcewrsyfq8fyx2gvs8s850smsgw0yq8ykfa8",
    "Ethereum address:": "0x29717BF51D8AFcA452459936d395668A576Bce66",
}


def run(*args):
    if not os.path.exists(CLI):
        pytest.skip("vgen-hip not built")
    p = subprocess.run([CLI, "verify", *args], capture_output=True, text=True, timeout=60)
    return p.returncode, p.stdout, p.stderr


@pytest.mark.parametrize("key, shown", [(WIF_U, WIF_U), (WIF_C, WIF_C), (HEX, WIF_C), ("0x" + HEX, WIF_C)])
def test_verify_lists_every_address_of_the_key(key, shown):
    rc, out, _ = run("-k", key)
    assert rc == 0
    lines = out.splitlines()
    assert lines[0] == f"Private key: {shown}"          # a WIF is echoed, a hex key shown as its compressed WIF
    assert lines[1].split() == ["WIF", "(uncompr.):", WIF_U]
    assert lines[2] == f"Hex: {HEX}" and lines[3] == ""
    for label, addr in ADDRS.items():
        assert any(ln.startswith(label) and ln.split()[-1] == addr for ln in lines), (label, out)


@pytest.mark.parametrize("address, verdict", [
    ("1GAehh7TsJAHuUAeKZcXf5CnwuGuGgyX2S", "MATCH!"),
    ("bc1qmy63mjadtw8nhzl69ukdepwzsyvv4yex5qlmkd".upper(), "MATCH!"),                      # BIP-173: one case, either
    ("Bc1qmy63mjadtw8nhzl69ukdepwzsyvv4yex5qlmkd", "MISMATCH! Expected: Bc1qmy63mjadtw8nhzl69ukdepwzsyvv4yex5qlmkd"),
    ("0x29717BF51D8AFcA452459936d395668A576Bce66", "MATCH!"),
    ("0x29717bf51d8afca452459936d395668a576bce66", "MATCH! (Ethereum, case-insensitive)"),
    ("29717bf51d8afca452459936d395668a576bce66", "MATCH! (Ethereum, case-insensitive)"),   # raw 40 hex
    ("1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH", "MISMATCH! Expected: 1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH"),
])
def test_verify_address_matching(address, verdict):
    rc, out, _ = run("-k", WIF_U, "-a", address)
    assert rc == 0 and out.splitlines()[-1] == verdict and out.splitlines()[-2] == ""


@pytest.mark.parametrize("key, message", [
    (WIF_C[:-1] + "8", "Invalid key format (not WIF or hex)"),   # checksum
    ("0c28", "Hex key must be 32 bytes"),
    ("zz", "Invalid key format (not WIF or hex)"),
    ("00" * 32, "malformed or out-of-range secret key"),
    ("ff" * 32, "malformed or out-of-range secret key"),
])
def test_verify_rejects_what_the_reference_rejects(key, message):
    rc, out, err = run("-k", key)
    assert rc == 1 and message in err and out == ""


@pytest.mark.parametrize("args", [["--key=" + HEX, "--address=1GAehh7TsJAHuUAeKZcXf5CnwuGuGgyX2S"], ["-k" + HEX, "-a1GAehh7TsJAHuUAeKZcXf5CnwuGuGgyX2S"],
                                  ["-k=" + WIF_U, "-a", "1GAehh7TsJAHuUAeKZcXf5CnwuGuGgyX2S"]])
def test_claps_spellings_of_an_argument(args):
    """--name=value, -nVALUE and -n=VALUE are the same argument to clap (the reference's parser) as -n VALUE."""
    rc, out, _ = run(*args)
    assert rc == 0 and out.splitlines()[-1] == "MATCH!"
