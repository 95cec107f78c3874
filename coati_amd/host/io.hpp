// Sequence I/O of the alignpair / sample verbs: FASTA, PHYLIP, JSON.
//
// Mirrors (format for format, including the quirks tests rely on):
//   extract_file_type                    src/lib/utils.cc:630-647
//   read_input / write_output            src/lib/io.cc:184-222,316-346
//   read_fasta / write_fasta             src/lib/fasta.cc:39-87,183-190
//   read_phylip / write_phylip           src/lib/phylip.cc:37-97,194-215
//   to_json/from_json/read/write_json    src/lib/json.cc:37-56,163-227
//   parse_matrix_csv (--sub)             src/lib/io.cc:48-88
#ifndef COATI_AMD_HOST_IO_HPP
#define COATI_AMD_HOST_IO_HPP

#include <iosfwd>
#include <string>

#include "model.hpp"
#include "seq.hpp"

namespace coati_amd {

struct file_type_t {
    std::string path;
    std::string type_ext;
};
// "file.ext" -> {file.ext, .ext};  "fmt:path" -> {path, .fmt} when the colon is past index 1
file_type_t extract_file_type(std::string path);

data_t read_fasta(std::istream& in);
data_t read_phylip(std::istream& in);
data_t read_json(std::istream& in);
void write_fasta(const data_t& data, std::ostream& out);
void write_phylip(const data_t& data, std::ostream& out);
void write_json(const data_t& data, std::ostream& out);
// element `iter` of a JSON array of `count` alignments (coati sample, and the batch extension)
void write_json(const data_t& data, std::ostream& out, std::size_t iter, std::size_t count);

// Dispatch on extension or "fmt:" prefix; empty path or "-" = stdin/stdout (JSON by default).
data_t read_input(const std::string& path);
void write_output(const data_t& data, const std::string& path);

// The score as nlohmann::json prints a float: widened to double, shortest round trip ("0.0").
std::string json_number(float value);

// `--sub file`: branch length on the first line, then "cod,cod,rate" lines -> exp(Q * t).
matrix61_t parse_matrix_csv(const std::string& path);

}  // namespace coati_amd
#endif
