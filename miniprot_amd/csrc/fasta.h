// fasta.h -- a small FASTA/FASTQ reader over zlib for the entry points that take file names (mp_idx_load on a FASTA
// genome, mp_map_file on a protein file).  Same record grammar as the reader the reference uses (kseq.h): a record starts
// at '>' or '@', the name ends at the first white space, the rest of the header line is a comment, sequence lines are
// concatenated (every byte but the newline and one trailing carriage return per line) until the next header or a '+' line; after '+' the quality string (as long as the
// sequence, possibly over several lines) is skipped.  "-" reads standard input.
#pragma once
#include <zlib.h>
#include <cctype>
#include <cstring>
#include <string>

namespace mpa {

class FastaReader {
public:
	explicit FastaReader(const char *fn) { fp_ = fn && strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(0, "r"); }
	~FastaReader() { if (fp_) gzclose(fp_); }
	bool ok() const { return fp_ != nullptr; }
	// false at the end of the file
	bool next(std::string &name, std::string &comment, std::string &seq)
	{
		int c;
		if (last_ == 0) {                                   // look for the first header
			while ((c = get()) >= 0 && c != '>' && c != '@') {}
			if (c < 0) return false;
			last_ = c;
		}
		name.clear(), comment.clear(), seq.clear();
		while ((c = get()) >= 0 && !isspace(c)) name.push_back((char)c);
		if (c >= 0 && c != '\n') {
			while ((c = get()) >= 0 && c != '\n') comment.push_back((char)c);
			while (!comment.empty() && comment.back() == '\r') comment.pop_back();
		}
		if (c < 0) { last_ = 0; at_end_ = true; return !name.empty(); }   // (a lone header character at the end of the file is no record, as in kseq_read)
		// sequence lines: a header character only counts at the start of a line.  Like kseq.h (the reader behind the reference's
		// mp_bseq_read / mp_ntseq_read) every byte of a line but its newline is sequence -- blanks and tabs inside a line stay and
		// later map to N / X -- and a line loses ONE trailing '\r' (unless that is all the sequence there is so far, kseq.h:146)
		bool bol = true;
		auto end_of_line = [&] { if (seq.size() > 1 && seq.back() == '\r') seq.pop_back(); };
		while ((c = get()) >= 0) {
			if (bol && (c == '>' || c == '@' || c == '+')) break;
			if (c == '\n') { if (!bol) end_of_line(); bol = true; continue; }
			bol = false;
			seq.push_back((char)c);
		}
		if (c < 0 && !bol) end_of_line();
		if (c == '>' || c == '@') { last_ = c; return true; }
		if (c == '+') {                                     // FASTQ: skip the '+' line and as many quality characters as bases
			while ((c = get()) >= 0 && c != '\n') {}
			size_t n = 0;
			while (n < seq.size() && (c = get()) >= 0) if (c != '\n' && c != '\r') ++n;
			last_ = 0;
			return true;
		}
		last_ = 0, at_end_ = true;
		return true;
	}
private:
	int get()
	{
		if (pos_ == len_) {
			if (at_end_) return -1;
			len_ = gzread(fp_, buf_, sizeof(buf_)), pos_ = 0;
			if (len_ <= 0) { len_ = 0, at_end_ = true; return -1; }
		}
		return (unsigned char)buf_[pos_++];
	}
	gzFile fp_ = nullptr;
	char buf_[1 << 16];
	int len_ = 0, pos_ = 0, last_ = 0;
	bool at_end_ = false;
};

} // namespace mpa
