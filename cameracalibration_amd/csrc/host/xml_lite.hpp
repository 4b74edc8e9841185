// xml_lite.hpp -- the small subset of XML the calibration config files use.
//
// The reference reads its config with Boost.PropertyTree's read_xml and then
// only calls get<std::string>(dotted.path, default) and iterates the children
// of data.camera.clouds (src/camera_calibration_optimization.cpp:150-158,207-209;
// :72-79).  This parser covers exactly that: elements, text, comments, the XML
// declaration; attributes are skipped; no entities beyond the five predefined.
#pragma once

#include <cctype>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ccal_host {

struct XmlNode {
    std::string name;
    std::string text;  // concatenated character data directly inside this element
    std::vector<std::unique_ptr<XmlNode>> children;

    const XmlNode* child(const std::string& n) const
    {
        for (const auto& c : children)
            if (c->name == n) return c.get();
        return nullptr;
    }

    // dotted path lookup, first match at every level (ptree::get semantics)
    const XmlNode* find(const std::string& path) const
    {
        const XmlNode* cur = this;
        size_t pos = 0;
        while (cur && pos <= path.size()) {
            size_t dot = path.find('.', pos);
            std::string key = path.substr(pos, dot == std::string::npos ? std::string::npos : dot - pos);
            cur = cur->child(key);
            if (dot == std::string::npos) break;
            pos = dot + 1;
        }
        return cur;
    }

    std::string get(const std::string& path, const std::string& dflt) const
    {
        const XmlNode* n = find(path);
        return n ? n->text : dflt;
    }
};

class XmlParser {
public:
    explicit XmlParser(const std::string& s) : s_(s) {}

    std::unique_ptr<XmlNode> parse()
    {
        auto root = std::make_unique<XmlNode>();
        skip_misc();
        while (pos_ < s_.size()) {
            if (s_[pos_] != '<') throw std::runtime_error("xml: text outside the root element");
            root->children.push_back(element());
            skip_misc();
        }
        return root;
    }

private:
    const std::string& s_;
    size_t pos_ = 0;

    bool starts(const char* lit) const { return s_.compare(pos_, std::char_traits<char>::length(lit), lit) == 0; }

    void skip_ws()
    {
        while (pos_ < s_.size() && std::isspace((unsigned char)s_[pos_])) ++pos_;
    }

    void skip_until(const char* lit)
    {
        size_t e = s_.find(lit, pos_);
        if (e == std::string::npos) throw std::runtime_error("xml: unterminated construct");
        pos_ = e + std::char_traits<char>::length(lit);
    }

    // whitespace, comments, processing instructions, doctype
    void skip_misc()
    {
        for (;;) {
            skip_ws();
            if (starts("<!--")) skip_until("-->");
            else if (starts("<?")) skip_until("?>");
            else if (starts("<!")) skip_until(">");
            else return;
        }
    }

    static void append_decoded(std::string& out, const std::string& raw)
    {
        for (size_t i = 0; i < raw.size(); ++i) {
            if (raw[i] == '&') {
                struct { const char* ent; char c; } tbl[] = {{"&lt;", '<'}, {"&gt;", '>'}, {"&amp;", '&'},
                                                            {"&quot;", '"'}, {"&apos;", '\''}};
                bool hit = false;
                for (auto& t : tbl) {
                    size_t L = std::char_traits<char>::length(t.ent);
                    if (raw.compare(i, L, t.ent) == 0) { out.push_back(t.c); i += L - 1; hit = true; break; }
                }
                if (hit) continue;
            }
            out.push_back(raw[i]);
        }
    }

    std::unique_ptr<XmlNode> element()
    {
        ++pos_;  // '<'
        size_t b = pos_;
        while (pos_ < s_.size() && !std::isspace((unsigned char)s_[pos_]) && s_[pos_] != '>' && s_[pos_] != '/') ++pos_;
        auto node = std::make_unique<XmlNode>();
        node->name = s_.substr(b, pos_ - b);
        if (node->name.empty()) throw std::runtime_error("xml: empty element name");
        // attributes are not used by the config schema: skip to the end of the tag
        bool self_closed = false;
        while (pos_ < s_.size() && s_[pos_] != '>') {
            if (s_[pos_] == '"' || s_[pos_] == '\'') {
                char q = s_[pos_++];
                while (pos_ < s_.size() && s_[pos_] != q) ++pos_;
            }
            if (s_[pos_] == '/' && pos_ + 1 < s_.size() && s_[pos_ + 1] == '>') self_closed = true;
            ++pos_;
        }
        if (pos_ >= s_.size()) throw std::runtime_error("xml: unterminated tag");
        ++pos_;  // '>'
        if (self_closed) return node;
        for (;;) {
            size_t lt = s_.find('<', pos_);
            if (lt == std::string::npos) throw std::runtime_error("xml: missing close tag for " + node->name);
            append_decoded(node->text, s_.substr(pos_, lt - pos_));
            pos_ = lt;
            if (starts("<!--")) { skip_until("-->"); continue; }
            if (starts("<![CDATA[")) {
                size_t e = s_.find("]]>", pos_);
                if (e == std::string::npos) throw std::runtime_error("xml: unterminated CDATA");
                node->text += s_.substr(pos_ + 9, e - pos_ - 9);
                pos_ = e + 3;
                continue;
            }
            if (starts("<?")) { skip_until("?>"); continue; }
            if (starts("</")) {
                size_t e = s_.find('>', pos_);
                if (e == std::string::npos) throw std::runtime_error("xml: unterminated close tag");
                std::string nm = s_.substr(pos_ + 2, e - pos_ - 2);
                while (!nm.empty() && std::isspace((unsigned char)nm.back())) nm.pop_back();
                if (nm != node->name) throw std::runtime_error("xml: mismatched close tag " + nm);
                pos_ = e + 1;
                break;
            }
            node->children.push_back(element());
        }
        // element-only content: drop the indentation whitespace
        bool all_ws = true;
        for (char c : node->text)
            if (!std::isspace((unsigned char)c)) { all_ws = false; break; }
        if (all_ws) node->text.clear();
        return node;
    }
};

inline std::string trim(const std::string& s)
{
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}

} // namespace ccal_host
