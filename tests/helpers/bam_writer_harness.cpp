// test harness: SAM text -> BAM through the CLI's BamWriter (merkurio_amd/csrc/cli/io.cpp) WITHOUT a device: records are
// encoded (BamWriter::encode_record), handed over in pieces of <piece> records (put_encoded: moved, never copied together),
// cut into BGZF members by the writer's queue + writer thread and deflated by zlib on the host threads (the --host-codec
// path; the device path shares everything but the deflate call).
// usage: harness <in.sam> <out.bam> <records per piece> [members per run of the writer queue, default 1536]
//        harness --members <header.sam> <out.bam> <file of finished BGZF members>...: the header through the writer, then members made
//        elsewhere passed through as they are (BamWriter::put_members: what `tag` does with the members the device hands back)
#include <cstdio>
#include <cstdlib>
#include <string>

#include "io.hpp"
using namespace cli;
int main(int argc, char **argv) {
    try {
        if (argc > 3 && std::string(argv[1]) == "--members") {
            SamFile h;
            h.open(argv[2]);
            BamWriter bw;
            bw.open(argv[3], h.header);
            for (int k = 4; k < argc; ++k) {
                FILE *g = fopen(argv[k], "rb");
                if (!g) bail(std::string("cannot open ") + argv[k]);
                fseek(g, 0, SEEK_END);
                const size_t n = (size_t)ftell(g);
                fseek(g, 0, SEEK_SET);
                BamWriter::RawBuffer b = bw.take_raw_buffer(n + (k % 2 ? 100000 : 0));  // (buffers larger than their content; recycled ones)
                if (fread(b.p, 1, n, g) != n) bail("short read");
                fclose(g);
                bw.put_members(b, n);
            }
            bw.close();
            printf("#members files %d\n", argc - 4);
            return 0;
        }
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
