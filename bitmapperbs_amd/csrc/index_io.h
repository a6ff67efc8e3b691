// bitmapperbs_amd/csrc/index_io.h -- pieces shared by the two index builders (index_io.cpp on the host cores,
// index_build_gpu.hip on the device): the FASTA reader, the in-memory image of the six index files and their writer.
#ifndef BMBS_INDEX_IO_H
#define BMBS_INDEX_IO_H
#include <cstdint>
#include <string>
#include <vector>

namespace bmbs_io {
typedef uint64_t u64; typedef uint32_t u32; typedef uint8_t u8;

struct Chrom { std::string name; u64 len; };

// what createIndex leaves on disk (Index.cpp:832-938), field for field
struct Built {
    std::vector<Chrom> chroms;
    u64 G = 0;
    std::vector<u8> pac;
    u64 sa_length = 0, shapline = 0, nacgt[5] = {0, 0, 0, 0, 0};
    std::vector<u64> bwt, high_occ, sa_flag;
    std::vector<u32> hash_hi, sa;
    std::vector<u8> hash_lo;
};

bool slurp_fasta(const char* path, std::vector<Chrom>& chroms, std::vector<char>& gen);
void prepare_genome(Built& B, std::vector<char>& gen, int n_threads);
// 16-mer table entries from the first/one-past-last row of every occurring 16-mer (top/bot; bot == 0: absent), bwt.cpp:1866-2010
void fill_hash_table(Built& B, const u64* top, const u64* bot);
int write_files(const Built& B, const std::string& base);
}  // namespace bmbs_io
#endif
