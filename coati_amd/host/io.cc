#include "io.hpp"

#include <algorithm>
#include <charconv>
#include <cctype>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "codon.hpp"

namespace coati_amd {

namespace {
std::string strip_spaces(std::string s) {
    s.erase(std::remove_if(s.begin(), s.end(), [](unsigned char c) { return std::isspace(c) != 0; }), s.end());
    return s;
}
std::string trim(const std::string& s) {
    std::size_t a = 0, b = s.size();
    while(a < b && std::isspace(static_cast<unsigned char>(s[a]))) ++a;
    while(b > a && std::isspace(static_cast<unsigned char>(s[b - 1]))) --b;
    return s.substr(a, b - a);
}
std::string extension_of(const std::string& path) {  // std::filesystem::path::extension semantics
    const std::size_t slash = path.find_last_of('/');
    const std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const std::size_t dot = name.find_last_of('.');
    if(dot == std::string::npos || dot == 0 || name == "..") return "";
    return name.substr(dot);
}

// ---- a minimal JSON reader: objects / arrays / strings / numbers / literals, key order kept ----
struct JsonReader {
    const std::string& s;
    std::size_t p{0};
    explicit JsonReader(const std::string& text) : s(text) {}
    [[noreturn]] void bad(const char* what) const { throw std::invalid_argument(std::string("Invalid JSON input: ") + what); }
    void ws() {
        while(p < s.size() && std::isspace(static_cast<unsigned char>(s[p]))) ++p;
    }
    char peek() {
        ws();
        if(p >= s.size()) bad("unexpected end");
        return s[p];
    }
    void expect(char c) {
        if(peek() != c) bad("unexpected character");
        ++p;
    }
    std::string string() {
        expect('"');
        std::string out;
        while(p < s.size() && s[p] != '"') {
            char c = s[p++];
            if(c == '\\') {
                if(p >= s.size()) bad("bad escape");
                const char e = s[p++];
                switch(e) {
                case 'n': out.push_back('\n'); break;
                case 't': out.push_back('\t'); break;
                case 'r': out.push_back('\r'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'u': {
                    if(p + 4 > s.size()) bad("bad \\u escape");
                    const unsigned cp = static_cast<unsigned>(std::stoul(s.substr(p, 4), nullptr, 16));
                    p += 4;
                    if(cp < 0x80) {
                        out.push_back(static_cast<char>(cp));
                    } else if(cp < 0x800) {
                        out.push_back(static_cast<char>(0xC0 | (cp >> 6)));
                        out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                    } else {
                        out.push_back(static_cast<char>(0xE0 | (cp >> 12)));
                        out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
                        out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                    }
                    break;
                }
                default: out.push_back(e);
                }
            } else {
                out.push_back(c);
            }
        }
        if(p >= s.size()) bad("unterminated string");
        ++p;
        return out;
    }
    double number() {
        ws();
        const std::size_t b = p;
        while(p < s.size() && (std::isdigit(static_cast<unsigned char>(s[p])) || std::strchr("+-.eE", s[p]) != nullptr)) ++p;
        if(b == p) bad("number expected");
        return std::stod(s.substr(b, p - b));
    }
    void skip_value() {
        const char c = peek();
        if(c == '"') {
            string();
        } else if(c == '{') {
            ++p;
            if(peek() == '}') { ++p; return; }
            for(;;) {
                string();
                expect(':');
                skip_value();
                if(peek() == ',') { ++p; continue; }
                expect('}');
                return;
            }
        } else if(c == '[') {
            ++p;
            if(peek() == ']') { ++p; return; }
            for(;;) {
                skip_value();
                if(peek() == ',') { ++p; continue; }
                expect(']');
                return;
            }
        } else if(std::isalpha(static_cast<unsigned char>(c))) {
            while(p < s.size() && std::isalpha(static_cast<unsigned char>(s[p]))) ++p;
        } else {
            number();
        }
    }
};

std::string json_escape(const std::string& s) {
    std::string out;
    for(const unsigned char c : s) {
        switch(c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\t': out += "\\t"; break;
        case '\r': out += "\\r"; break;
        case '\b': out += "\\b"; break;
        case '\f': out += "\\f"; break;
        default:
            if(c < 0x20) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "\\u%04x", c);
                out += buf;
            } else {
                out.push_back(static_cast<char>(c));
            }
        }
    }
    return out;
}

void write_json_object(const data_t& data, std::ostream& out) {
    out << "{\n  \"alignment\": {";
    for(std::size_t i = 0; i < data.size(); ++i)
        out << (i ? "," : "") << "\n    \"" << json_escape(data.names[i]) << "\": \"" << json_escape(data.seqs[i]) << "\"";
    out << (data.size() ? "\n  }" : "}") << ",\n  \"score\": " << json_number(data.score) << "\n}";
}
}  // namespace

file_type_t extract_file_type(std::string path) {
    path = trim(path);
    const std::size_t colon = path.find_first_of(':');
    if(colon != std::string::npos && colon > 1) return {path.substr(colon + 1), "." + path.substr(0, colon)};
    return {path, extension_of(path)};
}

data_t read_fasta(std::istream& in) {
    data_t fasta;
    std::string line, name, content;
    while(in.good()) {
        std::getline(in, line);
        if(line.empty() || line[0] == ';') continue;
        if(line[0] == '>') {
            if(!name.empty()) {
                fasta.seqs.push_back(content);
                name.clear();
            }
            name = line.substr(1);
            if(!name.empty() && name.back() == '\r') name.pop_back();
            if(name.empty()) throw std::invalid_argument("Input fasta file contains a sequence without a name.");
            fasta.names.push_back(name);
            content.clear();
        } else if(!name.empty()) {
            content += strip_spaces(line);
        }
    }
    if(!name.empty()) fasta.seqs.push_back(content);
    return fasta;
}

data_t read_phylip(std::istream& in) {
    data_t phylip;
    std::string tok, line;
    in >> tok;
    const int n_seqs = std::stoi(tok);
    in >> tok;
    (void)std::stoi(tok);  // declared length (not enforced upstream)
    if(n_seqs < 0) throw std::invalid_argument("Invalid phylip header.");
    phylip.names.resize(n_seqs);
    phylip.seqs.resize(n_seqs);
    for(int i = 0; i < n_seqs; ++i) {
        std::getline(in, line);
        if(line.empty()) std::getline(in, line);
        phylip.names[i] = strip_spaces(line.substr(0, 10));
        phylip.seqs[i] = line.size() > 10 ? strip_spaces(line.substr(10)) : std::string();
    }
    std::size_t count = 0;
    while(in.good() && n_seqs > 0) {
        std::getline(in, line);
        if(line.empty()) continue;
        phylip.seqs[count % n_seqs] += strip_spaces(line);
        ++count;
    }
    return phylip;
}

data_t read_json(std::istream& in) {
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    JsonReader r(text);
    data_t data;
    bool have_aln = false, have_score = false;
    r.expect('{');
    if(r.peek() != '}') {
        for(;;) {
            const std::string key = r.string();
            r.expect(':');
            if(key == "alignment") {
                have_aln = true;
                r.expect('{');
                if(r.peek() == '}') {
                    ++r.p;
                } else {
                    for(;;) {
                        data.names.push_back(r.string());
                        r.expect(':');
                        data.seqs.push_back(r.string());
                        if(r.peek() == ',') { ++r.p; continue; }
                        r.expect('}');
                        break;
                    }
                }
            } else if(key == "score") {
                have_score = true;
                data.score = static_cast<float>(r.number());
            } else {
                r.skip_value();
            }
            if(r.peek() == ',') { ++r.p; continue; }
            r.expect('}');
            break;
        }
    }
    if(!have_aln) throw std::invalid_argument("Invalid JSON input: key 'alignment' not found.");
    if(!have_score) throw std::invalid_argument("Invalid JSON input: key 'score' not found.");  // json.cc:50-56 (at())
    return data;
}

void write_fasta(const data_t& data, std::ostream& out) {
    for(std::size_t i = 0; i < data.size(); ++i) {
        out << ">" << data.names[i] << std::endl;
        for(std::size_t j = 0; j < data.seqs[i].size(); j += 60) out << data.seqs[i].substr(j, 60) << std::endl;
    }
}

void write_phylip(const data_t& data, std::ostream& out) {
    // (phylip.cc:194-215 reads seqs[0] unconditionally: undefined behaviour for an empty set)
    if(data.seqs.empty() || data.names.size() < data.seqs.size()) throw std::invalid_argument("PHYLIP output needs at least one named sequence.");
    out << data.size() << " " << data.seqs[0].length() << std::endl;
    std::size_t i = 50;
    for(std::size_t j = 0; j < data.size(); ++j) {
        std::string name = data.names[j].substr(0, 10);
        name.append(10 - name.length(), ' ');
        out << name << data.seqs[j].substr(0, i) << std::endl;
    }
    out << std::endl;
    for(; i < data.seqs[0].length(); i += 60) {
        for(std::size_t j = 0; j < data.size(); ++j) out << data.seqs[j].substr(i, 60) << std::endl;
        out << std::endl;
    }
}

std::string json_number(float value) {
    const double d = static_cast<double>(value);
    if(!std::isfinite(d)) return "null";  // what nlohmann::json dumps for inf/nan
    char buf[40];
    auto res = std::to_chars(buf, buf + sizeof buf, d);
    std::string s(buf, res.ptr);
    if(s.find_first_of(".e") == std::string::npos) s += ".0";
    return s;
}

void write_json(const data_t& data, std::ostream& out) {
    write_json_object(data, out);
    out << std::endl;
}

void write_json(const data_t& data, std::ostream& out, std::size_t iter, std::size_t count) {
    if(iter == 0) out << "[" << std::endl;
    write_json_object(data, out);
    if(iter + 1 < count)
        out << "," << std::endl;
    else
        out << std::endl << "]" << std::endl;
}

data_t read_input(const std::string& path) {
    file_type_t type = path.empty() ? file_type_t{"-", ".json"} : extract_file_type(path);
    std::ifstream file;
    std::istream* in = &std::cin;
    if(!(type.path.empty() || type.path == "-")) {
        file.open(type.path);  // (upstream opens the unstripped "fmt:path" string, io.cc:199; the stripped one is a superset)
        if(!file) throw std::invalid_argument("Opening input file " + path + " failed.");
        in = &file;
    }
    data_t data;
    if(type.type_ext == ".fa" || type.type_ext == ".fasta")
        data = read_fasta(*in);
    else if(type.type_ext == ".phy")
        data = read_phylip(*in);
    else if(type.type_ext == ".json")
        data = read_json(*in);
    else
        throw std::invalid_argument("Invalid input " + path + ".");
    data.path = path;
    return data;
}

void write_output(const data_t& data, const std::string& path) {
    file_type_t type = path.empty() ? file_type_t{"-", ".json"} : extract_file_type(path);
    std::ofstream file;
    std::ostream* out = &std::cout;
    if(type.path != "-") {
        file.open(type.path);
        if(!file) throw std::invalid_argument("Opening output file " + path + " failed.");
        out = &file;
    }
    if(type.type_ext == ".fa" || type.type_ext == ".fasta")
        write_fasta(data, *out);
    else if(type.type_ext == ".phy")
        write_phylip(data, *out);
    else if(type.type_ext == ".json")
        write_json(data, *out);
    else
        throw std::invalid_argument("Invalid output format " + type.type_ext + ".");
}

matrix61_t parse_matrix_csv(const std::string& path) {
    std::ifstream in(path);
    if(!in) throw std::invalid_argument("Opening input rate matrix file failed.");
    std::string line;
    if(!std::getline(in, line)) throw std::invalid_argument("Rate matrix file is empty.");
    const float br_len = std::stof(line);
    matrix61_t Q(61 * 61, 0.0f);
    std::size_t count = 0;
    while(std::getline(in, line)) {
        if(trim(line).empty()) continue;
        std::stringstream ss(line);
        std::string c1, c2, val;
        std::getline(ss, c1, ',');
        std::getline(ss, c2, ',');
        std::getline(ss, val);
        const int i = cod_int(trim(c1)), j = cod_int(trim(c2));
        if(i < 0 || j < 0 || is_stop64(i) || is_stop64(j)) throw std::invalid_argument("Invalid codon in rate matrix file.");
        Q[cod64_to_61(i) * 61 + cod64_to_61(j)] = std::stof(val);
        ++count;
    }
    if(count != 61 * 61) throw std::invalid_argument("Error reading substitution rate CSV file. Exiting!");
    return rate_matrix_p(Q, br_len);
}

}  // namespace coati_amd
