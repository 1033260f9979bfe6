// json_min.h -- a small strict JSON reader for the scene files (objects, arrays, strings, numbers, literals).
// The reference reads scenes with the vendored nlohmann/json single header (raytracer.cpp:18,132); only the
// subset of behaviour the loader relies on is reproduced here:
//   * numbers converted to float go through double first (nlohmann stores double / int64 and static_casts),
//     which for every literal a scene file can hold equals (float)strtod(text);
//   * looking up a missing key yields a null value (non-const operator[] inserts null);
//   * a malformed document throws (std::runtime_error instead of nlohmann::detail::parse_error).
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rtjson {

struct Value {
	enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
	bool b = false;
	double num = 0;
	std::string str;
	std::vector<Value> arr;
	std::map<std::string, Value> obj;

	bool has(const std::string& k) const { return kind == Object && obj.find(k) != obj.end(); }
	const Value& operator[](const std::string& k) const {
		static const Value nullv;
		if (kind != Object) return nullv;
		auto it = obj.find(k);
		return it == obj.end() ? nullv : it->second;
	}
	const Value& operator[](size_t i) const {
		if (kind != Array || i >= arr.size()) throw std::runtime_error("json: array index out of range");
		return arr[i];
	}
	size_t size() const { return kind == Array ? arr.size() : kind == Object ? obj.size() : 0; }
	bool is(const char* s) const { return kind == String && str == s; }
	double number() const {
		if (kind != Number) throw std::runtime_error("json: value is not a number");
		return num;
	}
	float f32() const { return (float)number(); }
	int i32() const { return (int)number(); }
	const std::string& string() const {
		if (kind != String) throw std::runtime_error("json: value is not a string");
		return str;
	}
};

class Parser {
	const char* p; const char* end;
	[[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("json parse error: ") + what); }
	void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
	bool lit(const char* s) { size_t n = strlen(s); if ((size_t)(end - p) >= n && !memcmp(p, s, n)) { p += n; return true; } return false; }
	std::string parseString() {
		std::string out;
		p++;   // opening quote
		while (true) {
			if (p >= end) fail("unterminated string");
			char c = *p++;
			if (c == '"') break;
			if (c == '\\') {
				if (p >= end) fail("bad escape");
				char e = *p++;
				switch (e) {
				case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
				case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
				case 'r': out += '\r'; break; case 't': out += '\t'; break;
				case 'u': {
					if (end - p < 4) fail("bad \\u escape");
					unsigned cp = (unsigned)strtoul(std::string(p, p + 4).c_str(), nullptr, 16); p += 4;
					if (cp < 0x80) out += (char)cp;
					else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
					else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
					break; }
				default: fail("bad escape");
				}
			} else out += c;
		}
		return out;
	}
	Value parseValue(int depth) {
		if (depth > 256) fail("nesting too deep");
		ws();
		if (p >= end) fail("unexpected end");
		Value v;
		char c = *p;
		if (c == '{') {
			v.kind = Value::Object; p++; ws();
			if (p < end && *p == '}') { p++; return v; }
			while (true) {
				ws();
				if (p >= end || *p != '"') fail("expected key");
				std::string k = parseString();
				ws();
				if (p >= end || *p != ':') fail("expected ':'");
				p++;
				v.obj[k] = parseValue(depth + 1);
				ws();
				if (p < end && *p == ',') { p++; continue; }
				if (p < end && *p == '}') { p++; break; }
				fail("expected ',' or '}'");
			}
		} else if (c == '[') {
			v.kind = Value::Array; p++; ws();
			if (p < end && *p == ']') { p++; return v; }
			while (true) {
				v.arr.push_back(parseValue(depth + 1));
				ws();
				if (p < end && *p == ',') { p++; continue; }
				if (p < end && *p == ']') { p++; break; }
				fail("expected ',' or ']'");
			}
		} else if (c == '"') { v.kind = Value::String; v.str = parseString(); }
		else if (lit("true")) { v.kind = Value::Bool; v.b = true; }
		else if (lit("false")) { v.kind = Value::Bool; v.b = false; }
		else if (lit("null")) { v.kind = Value::Null; }
		else if (c == '-' || (c >= '0' && c <= '9')) {
			const char* q = p;
			if (*q == '-') q++;
			if (q >= end || *q < '0' || *q > '9') fail("bad number");
			while (q < end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) q++;
			std::string tok(p, q);
			char* e = nullptr;
			v.kind = Value::Number; v.num = strtod(tok.c_str(), &e);
			if (e == tok.c_str() || *e) fail("bad number");
			p = q;
		} else fail("unexpected character");
		return v;
	}
public:
	static Value parse(const std::string& text) {
		Parser ps; ps.p = text.data(); ps.end = text.data() + text.size();
		Value v = ps.parseValue(0);
		ps.ws();
		if (ps.p != ps.end) ps.fail("trailing characters");
		return v;
	}
};

}  // namespace rtjson
