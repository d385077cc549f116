// provider.cpp — see provider.h.
#include "provider.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../device/device_types.h"

namespace vg {
namespace {

struct Known {
    const char *id;
    const char *address;
    int format;
};

// Target addresses pinned by the reference itself: puzzle 1 by src/provider.rs:75-87, puzzle 66 by
// README.md:102-108 / src/provider.rs:110.  Others come from the table file.
const Known KNOWN[] = {
    {"b1000/1", "1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH", VGF_P2PKH},
    {"b1000/66", "13zb1hQbWVsc2S7ZTZnP2G4undNNpdh5so", VGF_P2PKH},
};

int kind_format(const std::string &kind) {   // provider.rs:29-41
    if (kind == "p2pkh") return VGF_P2PKH;
    if (kind == "p2wpkh") return VGF_P2WPKH;
    if (kind == "p2tr") return VGF_P2TR;
    if (kind == "p2sh") return VGF_P2SH_P2WPKH;
    fprintf(stderr, "Warning: Unknown address kind '%s', defaulting to P2PKH\n", kind.c_str());
    return VGF_P2PKH;
}

bool hex_to_be32(const std::string &hex, uint8_t out[32]) {
    if (hex.empty() || hex.size() > 64) return false;
    memset(out, 0, 32);
    int nib = 0;
    for (size_t i = hex.size(); i-- > 0; nib++) {
        const char c = hex[i];
        int v = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
        if (v < 0) return false;
        out[31 - nib / 2] |= (uint8_t)(nib % 2 ? v << 4 : v);
    }
    return true;
}

// b1000 puzzle N: keys 2^(N-1) .. 2^N - 1
bool b1000_range(const std::string &id, ProviderResult &r) {
    if (id.compare(0, 6, "b1000/") != 0) return false;
    char *e;
    const long n = strtol(id.c_str() + 6, &e, 10);
    if (*e || e == id.c_str() + 6 || n < 1 || n > 256) return false;
    memset(r.start, 0, 32);
    memset(r.end, 0, 32);
    r.start[31 - (n - 1) / 8] = (uint8_t)(1u << ((n - 1) % 8));
    for (long b = 0; b < n; b++) r.end[31 - b / 8] |= (uint8_t)(1u << (b % 8));
    r.has_range = true;
    return true;
}

std::vector<std::string> split_csv(const std::string &line) {
    std::vector<std::string> f(1);
    for (char c : line) {
        if (c == ',') f.emplace_back();
        else if (c != '\r' && c != '\n' && c != ' ' && c != '\t') f.back().push_back(c);
    }
    return f;
}

// 1 found, 0 not in the file, -1 error
int lookup_file(const char *path, const std::string &id, ProviderResult &out, std::string &err) {
    FILE *f = fopen(path, "r");
    if (!f) {
        err = std::string("cannot open provider table '") + path + "'";
        return -1;
    }
    char line[1024];
    int found = 0, lineno = 0;
    while (!found && fgets(line, sizeof line, f)) {
        lineno++;
        if (line[0] == '#' || line[0] == '\n' || line[0] == '\r') continue;
        std::vector<std::string> c = split_csv(line);
        if (c.size() < 3 || c[1].empty()) {
            err = std::string("provider table '") + path + "': line " + std::to_string(lineno) + ": expected id,address,kind[,start,end]";
            found = -1;
            break;
        }
        std::string key = c[0];
        std::replace(key.begin(), key.end(), ':', '/');
        if (key != id) continue;
        out.address = c[1];
        out.format = (unsigned)kind_format(c[2]);
        out.has_range = false;
        if (c.size() >= 5 && !c[3].empty() && !c[4].empty()) {
            if (!hex_to_be32(c[3], out.start) || !hex_to_be32(c[4], out.end)) {
                err = std::string("provider table '") + path + "': line " + std::to_string(lineno) + ": bad key range";
                found = -1;
                break;
            }
            out.has_range = true;
        } else {
            b1000_range(id, out);
        }
        found = 1;
    }
    fclose(f);
    return found;
}

}  // namespace

int provider_resolve(const std::string &pattern, const char *table_path, ProviderResult &out, std::string &err) {
    const size_t colon = pattern.find(':');
    if (colon == std::string::npos) return 0;
    if (pattern.compare(0, colon, "boha") != 0) return 0;     // unknown provider name: an ordinary regex
    std::string id = pattern.substr(colon + 1);
    std::replace(id.begin(), id.end(), ':', '/');             // "b1000:66" and "b1000/66" name the same puzzle
    out = ProviderResult();
    if (table_path && *table_path) {
        const int r = lookup_file(table_path, id, out, err);
        if (r != 0) return r;
    }
    for (const Known &k : KNOWN)
        if (id == k.id) {
            out.address = k.address;
            out.format = (unsigned)k.format;
            b1000_range(id, out);
            return 1;
        }
    ProviderResult probe;
    if (b1000_range(id, probe))
        err = "Failed to get puzzle '" + id + "': its target address is not in the built-in table (the reference reads it "
              "from the boha crate); add a row 'id,address,kind' to a provider table file";
    else
        err = "Failed to get puzzle '" + id + "': unknown puzzle";
    return -1;
}

std::string regex_escape(const std::string &s) {
    std::string o;
    for (char c : s) {
        if (strchr("\\.+*?()|[]{}^$#&-~", c) && c) o.push_back('\\');
        o.push_back(c);
    }
    return o;
}

std::string provider_build_pattern(const std::string &address, size_t prefix_length) {
    return "^" + regex_escape(address.substr(0, std::min(prefix_length, address.size())));
}

std::string provider_build_exact_pattern(const std::string &address) { return "^" + regex_escape(address) + "$"; }

}  // namespace vg
