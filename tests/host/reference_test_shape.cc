// The reference's own test file (zk_census_test.go), restated line for line over include/zkcensus_prover.hpp: the same environment variables and defaults
// (getEnvVars, :14-51), the same artifacts tree (./artifacts/<name>/<env>/<nLevels>/), the same three tests in the same order --
//   Test_genInputs   (:53-72)    internal.MockInputs(nLevels, 10)          -> inputs_example.json
//   Test_genProof    (:74-101)   prover.Prove(zkey, wasm, inputs), Bytes() -> proof.json, signals.json
//   Test_verifyProof (:103-124)  prover.ParseProof, proof.Verify(vkey)
// -- run from the directory that holds ./artifacts (tests/test_gpu_reference_test_shape.py builds that tree around the build's own test key: the real proving key is
// a blob the reference does not ship, .MISSING_LARGE_BLOBS:1-3).  circuit.wasm is read when it is there and left empty when not (the key's shape names the circuit).
// Exit code 0 = all three passed; a failure prints the test and the error, as `go test` would, and exits 1.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <map>
#include <string>
#include <vector>
#include "zkcensus_prover.hpp"

using namespace zkcensus;

struct Fatal : std::runtime_error { using std::runtime_error::runtime_error; };
struct EnvVars { std::string circuitName, environment; int nLevels, keySize, nPaddingLeafs; };

static bool atoi_ok(const char* s, int* out) {                      // strconv.Atoi: the whole string, or an error
    if (!s || !*s) return false;
    char* end = nullptr; const long v = strtol(s, &end, 10);
    if (*end) return false;
    *out = (int)v; return true;
}
static EnvVars getEnvVars() {
    EnvVars e{"zkCensus", "dev", 160, 20, 100};
    if (const char* v = getenv("CIRCUIT_NAME")) if (*v) e.circuitName = v;
    if (const char* v = getenv("ENVIRONMENT")) if (*v) e.environment = v;
    int n;
    if (atoi_ok(getenv("NLEVELS"), &n)) {
        if (n < 10) throw Fatal("the number of levels must be 10 at least to support the current key length");
        e.nLevels = n;
    }
    if (atoi_ok(getenv("KEYSIZE"), &n)) {
        if (n > e.nLevels / 8) throw Fatal("the key size can not be bigger than ceil(nLevels/8)");
        e.keySize = n;
    }
    if (atoi_ok(getenv("PADDING"), &n)) e.nPaddingLeafs = n;
    return e;
}
static ByteSlice ReadFile(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Fatal("open " + path + ": no such file or directory");
    return ByteSlice(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}
static void WriteFile(const std::string& path, const ByteSlice& data) {
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f.write(data.data(), (std::streamsize)data.size())) throw Fatal("write " + path + " failed");
}
static std::string basePathOf(const EnvVars& e) { return "./artifacts/" + e.circuitName + "/" + e.environment + "/" + std::to_string(e.nLevels); }

static void Test_genInputs() {
    puts("Generating example of circuits inputs...");
    const EnvVars e = getEnvVars();
    printf("Config loaded:map[env:%s nLevels:%d keySize:%d nPaddingLeafs:%d name:%s]\n", e.environment.c_str(), e.nLevels, e.keySize, e.nPaddingLeafs, e.circuitName.c_str());
    const internal::circuitInputs inputs = internal::MockInputs(e.nLevels, 10);
    WriteFile(basePathOf(e) + "/inputs_example.json", inputs.Bytes());
}
static void Test_genProof() {
    puts("Generating proof for the circuit...");
    const EnvVars e = getEnvVars();
    const std::string basePath = basePathOf(e);
    // Get files
    const ByteSlice zkey = ReadFile(basePath + "/proving_key.zkey");
    ByteSlice wasm;
    try { wasm = ReadFile(basePath + "/circuit.wasm"); } catch (const Fatal&) {}
    const ByteSlice inputs = ReadFile(basePath + "/inputs_example.json");
    // Generate the proof
    const prover::Proof proof = prover::Prove(zkey, wasm, inputs);
    // Encode proof and public signals
    const auto enc = proof.Bytes();
    // Write encoded proof and public signals
    WriteFile(basePath + "/proof.json", enc.first);
    WriteFile(basePath + "/signals.json", enc.second);
}
static void Test_verifyProof() {
    puts("Verifiying proof of the circuit...");
    const EnvVars e = getEnvVars();
    const std::string basePath = basePathOf(e);
    // Get files
    const ByteSlice vkey = ReadFile(basePath + "/verification_key.json");
    const ByteSlice proofData = ReadFile(basePath + "/proof.json");
    const ByteSlice pubSignals = ReadFile(basePath + "/signals.json");
    // Parse proof
    const prover::Proof proof = prover::ParseProof(proofData, pubSignals);
    // Verify proof
    proof.Verify(vkey);
}

// not in the reference: proof.json / signals.json -> ParseProof -> Bytes() must give the files back byte for byte (run by name only)
static void Test_bytesRoundTrip() {
    const std::string basePath = basePathOf(getEnvVars());
    const ByteSlice proofData = ReadFile(basePath + "/proof.json"), pubSignals = ReadFile(basePath + "/signals.json");
    const auto enc = prover::ParseProof(proofData, pubSignals).Bytes();
    if (enc.first != proofData) throw Fatal("proof.json does not survive ParseProof -> Bytes");
    if (enc.second != pubSignals) throw Fatal("signals.json does not survive ParseProof -> Bytes");
}

// not in the reference: the encodings of internal/helpers.go:16-34 on the raw client values of ts_inputs/src/example.ts:340-346, printed as JSON for the test to compare
// with the reference's inputs_example.json (the same voter)
static std::vector<uint8_t> unhex(const std::string& h) {
    std::vector<uint8_t> out; for (size_t i = 0; i + 1 < h.size(); i += 2) out.push_back((uint8_t)strtol(h.substr(i, 2).c_str(), nullptr, 16)); return out;
}
static void Print_encodings() {
    const auto eid = unhex("7faeab7a7d250527d614e952ae8e446825bd1124c6def410844c7c383d1519a6"), addr = unhex("032234DBb3B6dA8c11DDdc26338867C769e66B00"),
               pw = unhex("70617373776f7264313233"),
               sig = unhex("7b6cac3c3b64d0b7fc10f0f6d4b8baf2548f246a748d25d8825becc1e2fa3c6e0a2654b042be487f0a352bc2c0577cde1440373197b4d93e09fc7502b61e9632");
    const auto e = internal::BytesToArbo(eid.data(), eid.size());
    const uint8_t weight[1] = {10}; const auto vh = internal::BytesToArbo(weight, 1);
    Big a{}; memcpy(a.data(), addr.data(), addr.size());
    printf("{\"electionId\":[\"%s\",\"%s\"],\"voteHash\":[\"%s\",\"%s\"],\"address\":\"%s\",\"password\":\"%s\",\"signature\":\"%s\"}\n",
           detail::dec(e[0].data()).c_str(), detail::dec(e[1].data()).c_str(), detail::dec(vh[0].data()).c_str(), detail::dec(vh[1].data()).c_str(), detail::dec(a.data()).c_str(),
           detail::dec(internal::BigToFF(pw.data(), pw.size()).data()).c_str(), detail::dec(internal::BigToFF(sig.data(), sig.size()).data()).c_str());
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "Print_encodings") { Print_encodings(); return 0; }
    // `go test -run <name>`: one test by name, or all three in file order
    const std::map<std::string, void (*)()> tests = {{"Test_genInputs", Test_genInputs}, {"Test_genProof", Test_genProof}, {"Test_verifyProof", Test_verifyProof},
                                                       {"Test_bytesRoundTrip", Test_bytesRoundTrip}};
    const char* order[3] = {"Test_genInputs", "Test_genProof", "Test_verifyProof"};
    int failed = 0;
    if (argc > 1 && !tests.count(argv[1])) { printf("testing: warning: no tests to run\n"); return 1; }
    for (const char* name : order) {
        if (argc > 1) { if (name != order[0]) break; name = argv[1]; }
        try { tests.at(name)(); printf("--- PASS: %s\n", name); }
        catch (const std::exception& ex) { printf("--- FAIL: %s\n    %s\n", name, ex.what()); failed++; }
    }
    puts(failed ? "FAIL" : "PASS");
    return failed ? 1 : 0;
}
