// test harness: SAM text -> BAM through the CLI's BamWriter (merkurio_amd/csrc/cli/io.cpp) WITHOUT a device: records are
// encoded (BamWriter::encode_record), handed over in pieces of <piece> records (put_encoded: moved, never copied together),
// cut into BGZF members by the writer's queue + writer thread and deflated by zlib on the host threads (the --host-codec
// path; the device path shares everything but the deflate call).
// usage: harness <in.sam> <out.bam> <records per piece> [members per run of the writer queue, default 1536]
#include <cstdio>
#include <cstdlib>

#include "io.hpp"
using namespace cli;
int main(int argc, char **argv) {
    try {
        SamFile f;
        f.open(argv[1]);
        const size_t per = (size_t)strtoull(argv[3], nullptr, 10);
        BamWriter bw;
        if (argc > 4) bw.run_members = (size_t)strtoull(argv[4], nullptr, 10);
        bw.open(argv[2], f.header);
        size_t total = 0;
        while (f.fill(1u << 20)) {
            std::vector<uint8_t> piece;
            size_t in_piece = 0;
            for (size_t i = 0; i < f.recs.size(); ++i) {
                std::string line;
                f.append_line(i, line);
                if (per == 0) {
                    bw.write_record(line);  // the one-record entry: small writes share an open piece
                } else {
                    bw.encode_record(line, piece);
                    if (++in_piece == per) {
                        bw.put_encoded(std::move(piece));
                        piece = std::vector<uint8_t>();
                        in_piece = 0;
                    }
                }
            }
            if (!piece.empty()) bw.put_encoded(std::move(piece));
            total += f.recs.size();
        }
        bw.close();
        printf("#records %zu\n", total);
    } catch (const Error &e) {
        printf("#error %s\n", e.what());
    }
    return 0;
}
