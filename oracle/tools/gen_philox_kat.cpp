// ORACLE tooling (test infrastructure): prints known-answer vectors of rocRAND's
// own Philox4x32-10 engine, evaluated on the HOST (the engine in
// rocrand_philox4x32_10.h is __host__ __device__), as JSON.  Used once, here, to
// pin oracle/philox_ref.py; output committed as tests/golden/philox_kat.json.
// Build+run: oracle/tools/gen_philox_kat.sh
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <cstdio>
#include <cstdint>

int main() {
    struct Q { unsigned long long seed, subseq, offset; };
    const Q qs[] = {
        {0ull, 0ull, 0ull},
        {42ull, 0ull, 0ull},
        {42ull, 7ull, 0ull},
        {42ull, 7ull, 4ull},
        {42ull, 8191ull, 4ull * ((123ull << 16) | 8ull)},
        {0xdeadbeefdeadbeefull, 0xffffffffffffffffull, 4ull * ((5ull << 16) | 1ull)},
        {0x0123456789abcdefull, 0x100000000ull, 4ull * 0x100000000ull},
        {1ull, 65535ull, 4ull * ((1000000ull << 16) | 107ull)},
    };
    printf("{\n \"source\": \"rocRAND %d host engine (rocrand_init + rocrand4)\",\n \"rocrand\": [\n", ROCRAND_VERSION);
    const int n = sizeof(qs) / sizeof(qs[0]);
    for (int i = 0; i < n; ++i) {
        rocrand_state_philox4x32_10 s;
        rocrand_init(qs[i].seed, qs[i].subseq, qs[i].offset, &s);
        uint4 r = rocrand4(&s);
        printf("  {\"seed\": \"%llu\", \"subseq\": \"%llu\", \"offset\": \"%llu\", \"out\": [%u, %u, %u, %u]}%s\n",
               qs[i].seed, qs[i].subseq, qs[i].offset, r.x, r.y, r.z, r.w, i + 1 < n ? "," : "");
    }
    printf(" ]\n}\n");
    return 0;
}
