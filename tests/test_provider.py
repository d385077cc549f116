"""Provider patterns: vgen_provider_resolve / vgen_provider_build_pattern against the
cases of the reference's own tests (src/provider.rs:65-150) and README (puzzle 66), plus the table file."""
import os
import subprocess

import pytest

import vgen_amd as vg


def test_reference_cases():
    assert vg.provider_resolve("^1Cat") is None                                   # provider.rs:69-72
    r = vg.provider_resolve("boha:b1000:1")                                       # provider.rs:75-87
    assert r.address == "1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH" and r.format == vg.AddressFormat.P2pkh
    assert r.key_range == (1, 1)
    assert vg.provider_resolve("boha:b1000/1").address == r.address               # provider.rs:90-100
    with pytest.raises(vg.VgenError):                                             # provider.rs:103-106
        vg.provider_resolve("boha:invalid:999999")
    p = vg.ProviderResult("13zb1hQbWVsc2S7ZTZnP2G4undNNpdh5so", vg.AddressFormat.P2pkh)
    assert vg.build_pattern(p, 6) == "^13zb1h" and vg.build_pattern(p, 10) == "^13zb1hQbWV"   # provider.rs:109-119
    assert vg.build_pattern(vg.ProviderResult("1Cat", vg.AddressFormat.P2pkh), 100) == "^1Cat"  # provider.rs:122-131
    assert vg.build_exact_pattern(p) == "^13zb1hQbWVsc2S7ZTZnP2G4undNNpdh5so$"     # provider.rs:134-146
    with pytest.raises(ValueError):
        vg.build_pattern(p, 0)                                                    # lib.rs:570-572


def test_puzzle_66_and_unknown_names():
    r = vg.provider_resolve("boha:b1000:66")                                      # README.md:102-108
    assert r.address == "13zb1hQbWVsc2S7ZTZnP2G4undNNpdh5so" and r.key_range == (2**65, 2**66 - 1)
    assert vg.provider_resolve("other:thing") is None                             # unknown provider = regex (provider.rs:19)
    assert vg.provider_resolve("a:b|c") is None
    with pytest.raises(vg.VgenError) as e:
        vg.provider_resolve("boha:b1000:161")
    assert "puzzles 1..160" in str(e.value)
    with pytest.raises(vg.VgenError) as e:
        vg.provider_resolve("boha:gsmg:1")
    assert "not in the built-in b1000 table" in str(e.value)
    # escaping of metacharacters (regex::escape)
    assert vg.build_exact_pattern(vg.ProviderResult("a.b+c", vg.AddressFormat.P2pkh)) == "^a\\.b\\+c$"


def test_table_file(tmp_path):
    t = tmp_path / "puzzles.csv"
    t.write_text("# id,address,kind,start,end\n"
                 "b1000/67,1BY8GQbnueYofwSuFAT3USAhGjPrkxDdW9,p2pkh\n"
                 "b1000:1,1OverrideXXXXXXXXXXXXXXXXXXXXXXXXX,p2pkh,5,ff\n"
                 "gsmg/1,bc1qexampleexampleexampleexampleexamplexx,p2wpkh\n"
                 "x/2,bc1pzzz,p2tr,,\n"
                 "x/3,3abc,p2sh,10,20\n")
    r = vg.provider_resolve("boha:b1000:67", str(t))
    assert r.address == "1BY8GQbnueYofwSuFAT3USAhGjPrkxDdW9" and r.key_range == (2**66, 2**67 - 1)
    r = vg.provider_resolve("boha:b1000/1", str(t))                               # file rows take precedence
    assert r.address.startswith("1Override") and r.key_range == (5, 255)
    r = vg.provider_resolve("boha:gsmg:1", str(t))
    assert r.format == vg.AddressFormat.P2wpkh and r.key_range is None
    assert vg.provider_resolve("boha:x/2", str(t)).format == vg.AddressFormat.P2tr
    r = vg.provider_resolve("boha:x/3", str(t))
    assert r.format == vg.AddressFormat.P2shP2wpkh and r.key_range == (16, 32)
    assert vg.provider_resolve("boha:b1000:66", str(t)).address == "13zb1hQbWVsc2S7ZTZnP2G4undNNpdh5so"   # falls through
    with pytest.raises(vg.VgenError):
        vg.provider_resolve("boha:b1000:66", str(tmp_path / "missing.csv"))
    bad = tmp_path / "bad.csv"
    bad.write_text("b1000/5,onlytwo\n")
    with pytest.raises(vg.VgenError):
        vg.provider_resolve("boha:b1000:5", str(bad))


def test_cli_messages(tmp_path):
    exe = os.path.join(os.path.dirname(vg.library_path()), "vgen-hip")
    run = lambda *a: subprocess.run([exe, *a], capture_output=True, text=True, timeout=60)
    out = run("range", "-p", "boha:b1000:66")
    assert "Provider: boha:b1000:66 → 13zb1hQbWVsc2S7ZTZnP2G4undNNpdh5so → exact match" in out.stderr      # lib.rs:611-615
    out = run("generate", "-p", "boha:b1000:1", "-l", "4")
    assert "Provider: boha:b1000:1 → 1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH → pattern '^1BgG'" in out.stderr    # lib.rs:578-581
    out = run("generate", "-p", "boha:b1000:1", "-l", "0")
    assert out.returncode == 1 and "--prefix-length must be at least 1 for provider patterns" in out.stderr
    out = run("generate", "-p", "^1Cat", "-l", "3")
    assert "Warning: --prefix-length is ignored for regex patterns" in out.stderr                           # lib.rs:585-587
    t = tmp_path / "t.csv"
    t.write_text("nokeys/1,1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH,p2pkh\n")
    out = run("range", "-p", "boha:nokeys:1", "--provider-table", str(t))
    assert out.returncode == 1 and "has no key range. Use --range or --puzzle" in out.stderr               # lib.rs:625-630


def test_b1000_table_is_complete_and_every_solved_puzzle_rederives():
    """All 160 puzzles resolve (the reference reads them from the boha crate, src/provider.rs:23-53).  For the 79
    solved ones the address must be what the public key hashes to — checked with the oracle AND the product's
    host-side derivation; the other 81 addresses must at least be well-formed Base58Check P2PKH strings."""
    import hashlib
    import json
    from oracle import pyoracle as vo
    data = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "b1000_puzzles.json")))["puzzles"]
    assert [p["n"] for p in data] == list(range(1, 161))
    b58 = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"
    derived = 0
    for p in data:
        n = p["n"]
        r = vg.provider_resolve("boha:b1000:%d" % n)
        assert r.address == p["address"] and r.format == vg.AddressFormat.P2pkh and r.key_range == (2 ** (n - 1), 2 ** n - 1)
        v = 0
        for ch in p["address"]:
            v = v * 58 + b58.index(ch)
        raw = v.to_bytes(25, "big")
        assert raw[0] == 0 and hashlib.sha256(hashlib.sha256(raw[:21]).digest()).digest()[:4] == raw[21:], n
        if p["key_hex"]:
            k = int(p["key_hex"], 16)
            assert 2 ** (n - 1) <= k < 2 ** n
            assert vo.generate(0, k)["address"] == p["address"] == vg.derive(0, k).address, n
            derived += 1
    assert derived == 79
