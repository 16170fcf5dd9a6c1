// Host-side bounds of the descriptor compiler (dusp_amd/csrc/program.hpp), on the CPU: descriptors that ask for absurd
// sizes come back as an error string — nothing is allocated from an unchecked count, nothing throws.
#include <cstdio>
#include <string>
#include <vector>

#include "../../dusp_amd/csrc/program.hpp"

using namespace dusp;

static std::vector<double> header(double n_units, double n_rings) {
    return {kMagic, 1, 48000, 256, n_units, n_rings, 0, 0, 0, 0, 0, 0};
}

int main() {
    int bad = 0, cases = 0;
    auto expect_error = [&](const char *what, const std::vector<double> &words, const char *needle) {
        Program P;
        std::string err;
        bool threw = false, ok = false;
        try { ok = compile(words.data(), words.size(), P, err); } catch (...) { threw = true; }
        cases++;
        if (ok || threw || err.find(needle) == std::string::npos) {
            bad++;
            std::printf("FAIL %s: ok=%d threw=%d err=%s\n", what, (int)ok, (int)threw, err.c_str());
        }
    };
    {   // a CircleBuffer ring of 2^40 samples
        auto w = header(1, 1);
        w.insert(w.end(), {1, 1099511627776.0});
        w.insert(w.end(), {(double)OP_OSC, 1, 1, 1, IN_CONST, 1, 440, 0, 0});
        expect_error("ring of 2^40 samples", w, "bad ring length");
    }
    {   // a Delay whose maxDelay is 2^40
        auto w = header(2, 0);
        w.insert(w.end(), {(double)OP_OSC, 1, 1, 1, IN_CONST, 1, 440, 0, 0});
        w.insert(w.end(), {(double)OP_DELAY, 2, 1, 0, IN_CONNECT, 3, 0, 0, 0, IN_CONST, 1, 480, 1099511627776.0});
        w[7] = 1;
        expect_error("Delay with maxDelay 2^40", w, "bad Delay record");
    }
    {   // three Delays of 1e9 samples each: every ring is in range, their sum is not
        auto w = header(4, 0);
        w.insert(w.end(), {(double)OP_OSC, 1, 1, 1, IN_CONST, 1, 440, 0, 0});
        for (int k = 0; k < 3; k++) w.insert(w.end(), {(double)OP_DELAY, 2, 1, 0, IN_CONNECT, 3, (double)k, 0, 0, IN_CONST, 1, 480, 1e9});
        w[7] = 3;
        expect_error("3 x 1e9-sample delay lines", w, "more than 2^31 samples");
    }
    {   // a million units announced by a 21-word descriptor
        auto w = header(1 << 20, 0);
        w.insert(w.end(), {(double)OP_OSC, 1, 1, 1, IN_CONST, 1, 440, 0, 0});
        expect_error("2^20 units in 21 words", w, "too short for its unit");
    }
    {   // 65536 rings announced, none present
        auto w = header(1, 65536);
        w.insert(w.end(), {(double)OP_OSC, 1, 1, 1, IN_CONST, 1, 440, 0, 0});
        expect_error("2^16 rings in 21 words", w, "too short for its unit");
    }
    {   // sanity: the well-formed descriptor compiles
        auto w = header(1, 0);
        w.insert(w.end(), {(double)OP_OSC, 1, 1, 1, IN_CONST, 1, 440, 0, 0});
        Program P;
        std::string err;
        cases++;
        if (!compile(w.data(), w.size(), P, err) || P.ops.size() != 1) { bad++; std::printf("FAIL plain Osc: %s\n", err.c_str()); }
    }
    std::printf("{\"cases\": %d, \"bad\": %d}\n", cases, bad);
    return bad ? 1 : 0;
}
