"""tools/wrap_md.py file.md [width]: re-wrap the prose of a Markdown file at `width` columns (default 128) in place.  Table rows, headings,
fenced code and lines that are already short are left alone; list items keep their marker and get a hanging indent."""
import re, sys, textwrap


def wrap(text, width=128):
    out, fence = [], False
    for line in text.split('\n'):
        if line.lstrip().startswith('```'):
            fence = not fence
            out.append(line); continue
        if fence or len(line) <= width or line.lstrip().startswith('|') or line.startswith('#'):
            out.append(line); continue
        m = re.match(r'^(\s*)([-*+] |\d+[.)] )?', line)
        lead, marker = m.group(1), m.group(2) or ''
        body = line[len(lead) + len(marker):]
        out.extend(textwrap.wrap(body, width=width, initial_indent=lead + marker, subsequent_indent=lead + ' ' * len(marker),
                                 break_long_words=False, break_on_hyphens=False))
    return '\n'.join(out)


if __name__ == '__main__':
    p = sys.argv[1]
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    text = open(p).read()
    open(p, 'w').write(wrap(text, w))
