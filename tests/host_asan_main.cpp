#include <cstdio>
#include <cstdlib>
#include <vector>
#include "genie_internal.h"
using namespace genie;
int main() {
    for (int64_t n : {1, 5, 17, 1000, 70000}) {
        std::vector<uint8_t> codes((size_t)n);
        unsigned s = 12345u + (unsigned)n;
        for (auto &c : codes) { s = s * 1664525u + 1013904223u; c = (s >> 24) & 3; }
        if (n == 1000) for (size_t i = 0; i < 400; i++) codes[i] = i % 6 < 2 ? 3 : (i % 6 < 3 ? 0 : 1);   // repeats
        for (int K : {0, 3, 8, 15})
        for (int fmt : {0, 1, 2}) {                                  // automatic (compact), 32-byte entries, compact entries
            if (fmt == 2 && K != 8) continue;
            HostIndex *h = nullptr;
            int rc = build_host_index(codes.data(), n, nullptr, K, 7, n == 17 ? 9 : 0, fmt, &h);
            if (rc) { printf("n=%lld K=%d rc=%d\n", (long long)n, K, rc); continue; }
            if (K > 0 && n >= K) {
                int32_t ex[2] = {10, 100};
                double mean; int32_t worst;
                rc = train_rmi(*h, 2, ex, &mean, &worst);
                printf("n=%lld K=%d train rc=%d mean=%.2f worst=%d\n", (long long)n, K, rc, mean, worst);
            }
            BlobHeader hdr; fill_header(*h, &hdr);
            std::vector<uint8_t> blob((size_t)hdr.total_bytes);
            rc = serialize(*h, blob.data(), (int64_t)blob.size());
            printf("n=%lld K=%d serialize rc=%d bytes=%lld P2=%d\n", (long long)n, K, rc, (long long)hdr.total_bytes, hdr.P2);
            // the image validator on the image just written, whole and truncated (host pointer stands in for the device's)
            DevIndex dev;
            int ok = dev_index_from_header(hdr, blob.data(), (int64_t)blob.size(), &dev);
            int cut = dev_index_from_header(hdr, blob.data(), (int64_t)blob.size() - 1, &dev);
            if (ok != 0 || cut == 0) { printf("validator: ok=%d cut=%d\n", ok, cut); return 1; }
            // the image without the K-mer hash table
            BlobHeader slim; fill_header(*h, &slim, GENIE_IMAGE_NO_SEED_TABLE);
            std::vector<uint8_t> blob2((size_t)slim.total_bytes);
            rc = serialize(*h, blob2.data(), (int64_t)blob2.size(), GENIE_IMAGE_NO_SEED_TABLE);
            if (rc || slim.total_bytes > hdr.total_bytes || dev_index_from_header(slim, blob2.data(), (int64_t)blob2.size(), &dev)) { printf("slim image: rc=%d\n", rc); return 1; }
            delete h;
        }
    }
    return 0;
}
