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

// The b1000 collection ("Bitcoin puzzle transaction", 160 P2PKH outputs; puzzle N's key lies in 2^(N-1) .. 2^N - 1):
// what the reference obtains from the un-vendored `boha` crate (src/provider.rs:23-53).  The reference itself pins
// puzzle 1 (src/provider.rs:75-87) and puzzle 66 (README.md:102-108, src/provider.rs:110).  `derivable`: the puzzle
// has been solved and its key is public — tests/test_provider.py re-derives those 79 addresses from the keys
// (tests/golden/b1000_puzzles.json) with the oracle and with the product; the other 81 are public addresses pinned
// by their Base58Check checksum.
struct Puzzle {
    int n;
    const char *address;
    bool derivable;
};

const Puzzle B1000[] = {
    {1, "1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH", true},
    {2, "1CUNEBjYrCn2y1SdiUMohaKUi4wpP326Lb", true},
    {3, "19ZewH8Kk1PDbSNdJ97FP4EiCjTRaZMZQA", true},
    {4, "1EhqbyUMvvs7BfL8goY6qcPbD6YKfPqb7e", true},
    {5, "1E6NuFjCi27W5zoXg8TRdcSRq84zJeBW3k", true},
    {6, "1PitScNLyp2HCygzadCh7FveTnfmpPbfp8", true},
    {7, "1McVt1vMtCC7yn5b9wgX1833yCcLXzueeC", true},
    {8, "1M92tSqNmQLYw33fuBvjmeadirh1ysMBxK", true},
    {9, "1CQFwcjw1dwhtkVWBttNLDtqL7ivBonGPV", true},
    {10, "1LeBZP5QCwwgXRtmVUvTVrraqPUokyLHqe", true},
    {11, "1PgQVLmst3Z314JrQn5TNiys8Hc38TcXJu", true},
    {12, "1DBaumZxUkM4qMQRt2LVWyFJq5kDtSZQot", true},
    {13, "1Pie8JkxBT6MGPz9Nvi3fsPkr2D8q3GBc1", true},
    {14, "1ErZWg5cFCe4Vw5BzgfzB74VNLaXEiEkhk", true},
    {15, "1QCbW9HWnwQWiQqVo5exhAnmfqKRrCRsvW", true},
    {16, "1BDyrQ6WoF8VN3g9SAS1iKZcPzFfnDVieY", true},
    {17, "1HduPEXZRdG26SUT5Yk83mLkPyjnZuJ7Bm", true},
    {18, "1GnNTmTVLZiqQfLbAdp9DVdicEnB5GoERE", true},
    {19, "1NWmZRpHH4XSPwsW6dsS3nrNWfL1yrJj4w", true},
    {20, "1HsMJxNiV7TLxmoF6uJNkydxPFDog4NQum", true},
    {21, "14oFNXucftsHiUMY8uctg6N487riuyXs4h", true},
    {22, "1CfZWK1QTQE3eS9qn61dQjV89KDjZzfNcv", true},
    {23, "1L2GM8eE7mJWLdo3HZS6su1832NX2txaac", true},
    {24, "1rSnXMr63jdCuegJFuidJqWxUPV7AtUf7", true},
    {25, "15JhYXn6Mx3oF4Y7PcTAv2wVVAuCFFQNiP", true},
    {26, "1JVnST957hGztonaWK6FougdtjxzHzRMMg", true},
    {27, "128z5d7nN7PkCuX5qoA4Ys6pmxUYnEy86k", true},
    {28, "12jbtzBb54r97TCwW3G1gCFoumpckRAPdY", true},
    {29, "19EEC52krRUK1RkUAEZmQdjTyHT7Gp1TYT", true},
    {30, "1LHtnpd8nU5VHEMkG2TMYYNUjjLc992bps", true},
    {31, "1LhE6sCTuGae42Axu1L1ZB7L96yi9irEBE", true},
    {32, "1FRoHA9xewq7DjrZ1psWJVeTer8gHRqEvR", true},
    {33, "187swFMjz1G54ycVU56B7jZFHFTNVQFDiu", true},
    {34, "1PWABE7oUahG2AFFQhhvViQovnCr4rEv7Q", true},
    {35, "1PWCx5fovoEaoBowAvF5k91m2Xat9bMgwb", true},
    {36, "1Be2UF9NLfyLFbtm3TCbmuocc9N1Kduci1", true},
    {37, "14iXhn8bGajVWegZHJ18vJLHhntcpL4dex", true},
    {38, "1HBtApAFA9B2YZw3G2YKSMCtb3dVnjuNe2", true},
    {39, "122AJhKLEfkFBaGAd84pLp1kfE7xK3GdT8", true},
    {40, "1EeAxcprB2PpCnr34VfZdFrkUWuxyiNEFv", true},
    {41, "1L5sU9qvJeuwQUdt4y1eiLmquFxKjtHr3E", true},
    {42, "1E32GPWgDyeyQac4aJxm9HVoLrrEYPnM4N", true},
    {43, "1PiFuqGpG8yGM5v6rNHWS3TjsG6awgEGA1", true},
    {44, "1CkR2uS7LmFwc3T2jV8C1BhWb5mQaoxedF", true},
    {45, "1NtiLNGegHWE3Mp9g2JPkgx6wUg4TW7bbk", true},
    {46, "1F3JRMWudBaj48EhwcHDdpeuy2jwACNxjP", true},
    {47, "1Pd8VvT49sHKsmqrQiP61RsVwmXCZ6ay7Z", true},
    {48, "1DFYhaB2J9q1LLZJWKTnscPWos9VBqDHzv", true},
    {49, "12CiUhYVTTH33w3SPUBqcpMoqnApAV4WCF", true},
    {50, "1MEzite4ReNuWaL5Ds17ePKt2dCxWEofwk", true},
    {51, "1NpnQyZ7x24ud82b7WiRNvPm6N8bqGQnaS", true},
    {52, "15z9c9sVpu6fwNiK7dMAFgMYSK4GqsGZim", true},
    {53, "15K1YKJMiJ4fpesTVUcByoz334rHmknxmT", true},
    {54, "1KYUv7nSvXx4642TKeuC2SNdTk326uUpFy", true},
    {55, "1LzhS3k3e9Ub8i2W1V8xQFdB8n2MYCHPCa", true},
    {56, "17aPYR1m6pVAacXg1PTDDU7XafvK1dxvhi", true},
    {57, "15c9mPGLku1HuW9LRtBf4jcHVpBUt8txKz", true},
    {58, "1Dn8NF8qDyyfHMktmuoQLGyjWmZXgvosXf", true},
    {59, "1HAX2n9Uruu9YDt4cqRgYcvtGvZj1rbUyt", true},
    {60, "1Kn5h2qpgw9mWE5jKpk8PP4qvvJ1QVy8su", true},
    {61, "1AVJKwzs9AskraJLGHAZPiaZcrpDr1U6AB", true},
    {62, "1Me6EfpwZK5kQziBwBfvLiHjaPGxCKLoJi", true},
    {63, "1NpYjtLira16LfGbGwZJ5JbDPh3ai9bjf4", true},
    {64, "16jY7qLJnxb7CHZyqBP8qca9d51gAjyXQN", true},
    {65, "18ZMbwUFLMHoZBbfpCjUJQTCMCbktshgpe", true},
    {66, "13zb1hQbWVsc2S7ZTZnP2G4undNNpdh5so", true},
    {67, "1BY8GQbnueYofwSuFAT3USAhGjPrkxDdW9", true},
    {68, "1MVDYgVaSN6iKKEsbzRUAYFrYJadLYZvvZ", true},
    {69, "19vkiEajfhuZ8bs8Zu2jgmC6oqZbWqhxhG", true},
    {70, "19YZECXj3SxEZMoUeJ1yiPsw8xANe7M7QR", true},
    {71, "1PWo3JeB9jrGwfHDNpdGK54CRas7fsVzXU", false},
    {72, "1JTK7s9YVYywfm5XUH7RNhHJH1LshCaRFR", false},
    {73, "12VVRNPi4SJqUTsp6FmqDqY5sGosDtysn4", false},
    {74, "1FWGcVDK3JGzCC3WtkYetULPszMaK2Jksv", false},
    {75, "1J36UjUByGroXcCvmj13U6uwaVv9caEeAt", true},
    {76, "1DJh2eHFYQfACPmrvpyWc8MSTYKh7w9eRF", false},
    {77, "1Bxk4CQdqL9p22JEtDfdXMsng1XacifUtE", false},
    {78, "15qF6X51huDjqTmF9BJgxXdt1xcj46Jmhb", false},
    {79, "1ARk8HWJMn8js8tQmGUJeQHjSE7KRkn2t8", false},
    {80, "1BCf6rHUW6m3iH2ptsvnjgLruAiPQQepLe", true},
    {81, "15qsCm78whspNQFydGJQk5rexzxTQopnHZ", false},
    {82, "13zYrYhhJxp6Ui1VV7pqa5WDhNWM45ARAC", false},
    {83, "14MdEb4eFcT3MVG5sPFG4jGLuHJSnt1Dk2", false},
    {84, "1CMq3SvFcVEcpLMuuH8PUcNiqsK1oicG2D", false},
    {85, "1Kh22PvXERd2xpTQk3ur6pPEqFeckCJfAr", true},
    {86, "1K3x5L6G57Y494fDqBfrojD28UJv4s5JcK", false},
    {87, "1PxH3K1Shdjb7gSEoTX7UPDZ6SH4qGPrvq", false},
    {88, "16AbnZjZZipwHMkYKBSfswGWKDmXHjEpSf", false},
    {89, "19QciEHbGVNY4hrhfKXmcBBCrJSBZ6TaVt", false},
    {90, "1L12FHH2FHjvTviyanuiFVfmzCy46RRATU", true},
    {91, "1EzVHtmbN4fs4MiNk3ppEnKKhsmXYJ4s74", false},
    {92, "1AE8NzzgKE7Yhz7BWtAcAAxiFMbPo82NB5", false},
    {93, "17Q7tuG2JwFFU9rXVj3uZqRtioH3mx2Jad", false},
    {94, "1K6xGMUbs6ZTXBnhw1pippqwK6wjBWtNpL", false},
    {95, "19eVSDuizydXxhohGh8Ki9WY9KsHdSwoQC", true},
    {96, "15ANYzzCp5BFHcCnVFzXqyibpzgPLWaD8b", false},
    {97, "18ywPwj39nGjqBrQJSzZVq2izR12MDpDr8", false},
    {98, "1CaBVPrwUxbQYYswu32w7Mj4HR4maNoJSX", false},
    {99, "1JWnE6p6UN7ZJBN7TtcbNDoRcjFtuDWoNL", false},
    {100, "1KCgMv8fo2TPBpddVi9jqmMmcne9uSNJ5F", true},
    {101, "1CKCVdbDJasYmhswB6HKZHEAnNaDpK7W4n", false},
    {102, "1PXv28YxmYMaB8zxrKeZBW8dt2HK7RkRPX", false},
    {103, "1AcAmB6jmtU6AiEcXkmiNE9TNVPsj9DULf", false},
    {104, "1EQJvpsmhazYCcKX5Au6AZmZKRnzarMVZu", false},
    {105, "1CMjscKB3QW7SDyQ4c3C3DEUHiHRhiZVib", true},
    {106, "18KsfuHuzQaBTNLASyj15hy4LuqPUo1FNB", false},
    {107, "15EJFC5ZTs9nhsdvSUeBXjLAuYq3SWaxTc", false},
    {108, "1HB1iKUqeffnVsvQsbpC6dNi1XKbyNuqao", false},
    {109, "1GvgAXVCbA8FBjXfWiAms4ytFeJcKsoyhL", false},
    {110, "12JzYkkN76xkwvcPT6AWKZtGX6w2LAgsJg", true},
    {111, "1824ZJQ7nKJ9QFTRBqn7z7dHV5EGpzUpH3", false},
    {112, "18A7NA9FTsnJxWgkoFfPAFbQzuQxpRtCos", false},
    {113, "1NeGn21dUDDeqFQ63xb2SpgUuXuBLA4WT4", false},
    {114, "174SNxfqpdMGYy5YQcfLbSTK3MRNZEePoy", false},
    {115, "1NLbHuJebVwUZ1XqDjsAyfTRUPwDQbemfv", true},
    {116, "1MnJ6hdhvK37VLmqcdEwqC3iFxyWH2PHUV", false},
    {117, "1KNRfGWw7Q9Rmwsc6NT5zsdvEb9M2Wkj5Z", false},
    {118, "1PJZPzvGX19a7twf5HyD2VvNiPdHLzm9F6", false},
    {119, "1GuBBhf61rnvRe4K8zu8vdQB3kHzwFqSy7", false},
    {120, "17s2b9ksz5y7abUm92cHwG8jEPCzK3dLnT", false},
    {121, "1GDSuiThEV64c166LUFC9uDcVdGjqkxKyh", false},
    {122, "1Me3ASYt5JCTAK2XaC32RMeH34PdprrfDx", false},
    {123, "1CdufMQL892A69KXgv6UNBD17ywWqYpKut", false},
    {124, "1BkkGsX9ZM6iwL3zbqs7HWBV7SvosR6m8N", false},
    {125, "1PXAyUB8ZoH3WD8n5zoAthYjN15yN5CVq5", false},
    {126, "1AWCLZAjKbV1P7AHvaPNCKiB7ZWVDMxFiz", false},
    {127, "1G6EFyBRU86sThN3SSt3GrHu1sA7w7nzi4", false},
    {128, "1MZ2L1gFrCtkkn6DnTT2e4PFUTHw9gNwaj", false},
    {129, "1Hz3uv3nNZzBVMXLGadCucgjiCs5W9vaGz", false},
    {130, "1Fo65aKq8s8iquMt6weF1rku1moWVEd5Ua", false},
    {131, "16zRPnT8znwq42q7XeMkZUhb1bKqgRogyy", false},
    {132, "1KrU4dHE5WrW8rhWDsTRjR21r8t3dsrS3R", false},
    {133, "17uDfp5r4n441xkgLFmhNoSW1KWp6xVLD", false},
    {134, "13A3JrvXmvg5w9XGvyyR4JEJqiLz8ZySY3", false},
    {135, "16RGFo6hjq9ym6Pj7N5H7L1NR1rVPJyw2v", false},
    {136, "1UDHPdovvR985NrWSkdWQDEQ1xuRiTALq", false},
    {137, "15nf31J46iLuK1ZkTnqHo7WgN5cARFK3RA", false},
    {138, "1Ab4vzG6wEQBDNQM1B2bvUz4fqXXdFk2WT", false},
    {139, "1Fz63c775VV9fNyj25d9Xfw3YHE6sKCxbt", false},
    {140, "1QKBaU6WAeycb3DbKbLBkX7vJiaS8r42Xo", false},
    {141, "1CD91Vm97mLQvXhrnoMChhJx4TP9MaQkJo", false},
    {142, "15MnK2jXPqTMURX4xC3h4mAZxyCcaWWEDD", false},
    {143, "13N66gCzWWHEZBxhVxG18P8wyjEWF9Yoi1", false},
    {144, "1NevxKDYuDcCh1ZMMi6ftmWwGrZKC6j7Ux", false},
    {145, "19GpszRNUej5yYqxXoLnbZWKew3KdVLkXg", false},
    {146, "1M7ipcdYHey2Y5RZM34MBbpugghmjaV89P", false},
    {147, "18aNhurEAJsw6BAgtANpexk5ob1aGTwSeL", false},
    {148, "1FwZXt6EpRT7Fkndzv6K4b4DFoT4trbMrV", false},
    {149, "1CXvTzR6qv8wJ7eprzUKeWxyGcHwDYP1i2", false},
    {150, "1MUJSJYtGPVGkBCTqGspnxyHahpt5Te8jy", false},
    {151, "13Q84TNNvgcL3HJiqQPvyBb9m4hxjS3jkV", false},
    {152, "1LuUHyrQr8PKSvbcY1v1PiuGuqFjWpDumN", false},
    {153, "18192XpzzdDi2K11QVHR7td2HcPS6Qs5vg", false},
    {154, "1NgVmsCCJaKLzGyKLFJfVequnFW9ZvnMLN", false},
    {155, "1AoeP37TmHdFh8uN72fu9AqgtLrUwcv2wJ", false},
    {156, "1FTpAbQa4h8trvhQXjXnmNhqdiGBd1oraE", false},
    {157, "14JHoRAdmJg3XR4RjMDh6Wed6ft6hzbQe9", false},
    {158, "19z6waranEf8CcP8FqNgdwUe1QRxvUNKBG", false},
    {159, "14u4nA5sugaswb6SZgn5av2vuChdMnD9E5", false},
    {160, "1NBC8uXJy1GiJ6drkiZa1WuKn51ps7EPTv", false},
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
    ProviderResult probe;
    if (b1000_range(id, probe)) {
        const long n = strtol(id.c_str() + 6, nullptr, 10);
        for (const Puzzle &pz : B1000)
            if (pz.n == n) {
                out = probe;
                out.address = pz.address;
                out.format = VGF_P2PKH;
                return 1;
            }
        err = "Failed to get puzzle '" + id + "': the b1000 collection has puzzles 1..160";
        return -1;
    }
    err = "Failed to get puzzle '" + id + "': not in the built-in b1000 table (the reference reads other collections from the "
          "boha crate); add a row 'id,address,kind[,start_hex,end_hex]' to a provider table file";
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
