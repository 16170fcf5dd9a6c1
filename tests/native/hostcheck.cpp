// The untrusted-input path of the library on the CPU, under AddressSanitizer and UBSan (make -C dusp_amd/csrc hostcheck): descriptor words ->
// program (program.hpp: parse, channel inference, expansion) -> plans (fused_plan.hpp: fused voice shapes, the wave engine) ->
// kernel text (jit_codegen.hpp jit_source_from_descriptor — the very function dusp_circuit_kernel_source runs in front of the run-time
// compiler; the text is generated, not compiled).  Driven with
//   * every descriptor file named on the command line (the golden descriptors the reference generated) as it stands, over a spread of
//     workgroup geometries and knob settings;
//   * its truncations (every third length) and single-word corruptions (NaN, Inf, negative, fractional, huge, small-integer values at random
//     positions: the corpus of tests/test_gpu_parity.py::test_malformed_descriptors_are_rejected_not_crashed, which needs a GPU).
// Every call must come back with a verdict — 0 text, 1 malformed, 2 unsupported — and a message; the sanitizers abort on anything else
// (-fno-sanitize-recover).  Prints one JSON line.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../dusp_amd/csrc/jit_codegen.hpp"
#include "../../dusp_amd/csrc/ring_windows.hpp"

using namespace dusp;

static long g_text = 0, g_malformed = 0, g_unsupported = 0, g_calls = 0, g_bad = 0;
static size_t g_text_bytes = 0;

static void drive(const std::vector<double> &words, bool every_geometry) {
    // the plans dusp_program_build consults before the circuit compiler
    {
        Program P;
        std::string err;
        if (compile(words.data(), words.size(), P, err)) {
            FusedPlan fp;
            (void)plan_fused(P, fp);
            WavePlan wp;
            (void)plan_wave(P, wp, false);
            std::vector<RingWindow> wins;
            size_t covered = 0;
            ring_windows(P, 40, wins, covered);
        } else if (err.empty()) {
            g_bad++;
            std::printf("FAIL: compile() refused a descriptor without a message\n");
        }
    }
    static const int geo[][2] = {{4, 1}, {16, 1}, {16, 2}, {8, 4}, {1, 1}};
    const int n_geo = every_geometry ? 5 : 2;
    for (int gi = 0; gi < n_geo; gi++)
        for (int variant = 0; variant < (every_geometry ? 4 : 2); variant++) {
            JitSourceRequest rq;
            rq.waves = geo[gi][0];
            rq.per_wave = geo[gi][1];
            rq.continued = variant == 1;
            rq.scan_knob = variant == 2 ? 0 : variant == 3 ? 2 : 1;
            rq.lean_recurrence = variant == 2;
            rq.lds_table = variant != 3;
            rq.delay_line = variant == 3;
            JitSource src;
            std::string err;
            const int v = jit_source_from_descriptor(words.data(), words.size(), rq, src, err);
            g_calls++;
            if (v == 0) {
                g_text++;
                g_text_bytes += src.text.size();
                if (src.text.find("dusp_jit_render") == std::string::npos) g_bad++, std::printf("FAIL: a text without a render kernel\n");
            } else if (v == 1) g_malformed++;
            else if (v == 2) g_unsupported++;
            else g_bad++, std::printf("FAIL: verdict %d\n", v);
            if (v != 0 && err.empty()) g_bad++, std::printf("FAIL: verdict %d without a message\n", v);
        }
}

int main(int argc, char **argv) {
    std::mt19937_64 rng(7);
    const double poison[] = {NAN, INFINITY, -INFINITY, -1.0, 0.5, 1e18, -1e18, 3.0, 65536.0, 1099511627776.0, 0.0, 1.0, 2.0, 255.0, 4294967296.0, -0.0, 1e-300, 7.0, 40.0};
    const int n_poison = (int)(sizeof poison / sizeof poison[0]);
    int files = 0;
    const int corruptions = argc > 1 && std::getenv("HOSTCHECK_CORRUPTIONS") ? std::atoi(std::getenv("HOSTCHECK_CORRUPTIONS")) : 60;
    for (int a = 1; a < argc; a++) {
        FILE *f = std::fopen(argv[a], "rb");
        if (!f) {
            std::printf("FAIL: cannot open %s\n", argv[a]);
            g_bad++;
            continue;
        }
        std::vector<double> words;
        double w;
        while (std::fread(&w, sizeof w, 1, f) == 1) words.push_back(w);
        std::fclose(f);
        files++;
        drive(words, true);
        for (size_t k = 0; k < words.size(); k += 3) drive(std::vector<double>(words.begin(), words.begin() + (long)k), false);
        for (int c = 0; c < corruptions && !words.empty(); c++) {
            std::vector<double> d2 = words;
            d2[(size_t)(rng() % d2.size())] = poison[rng() % (unsigned)n_poison];
            if (c % 5 == 4) d2[(size_t)(rng() % d2.size())] = poison[rng() % (unsigned)n_poison];  // (two at once)
            drive(d2, false);
        }
    }
    std::printf("{\"files\": %d, \"calls\": %ld, \"text\": %ld, \"malformed\": %ld, \"unsupported\": %ld, \"text_bytes\": %zu, \"bad\": %ld}\n", files, g_calls, g_text, g_malformed,
                g_unsupported, g_text_bytes, g_bad);
    return g_bad ? 1 : 0;
}
