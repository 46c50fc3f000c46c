"""tools/wrap_md.py file.md [width]: re-flow the prose of a Markdown file to `width` columns (default 128) in place.  Paragraphs and list
items (with their continuation lines) are joined and re-wrapped with a hanging indent; table rows, headings, fenced code and blank lines are
left alone."""
import re, sys, textwrap

ITEM = re.compile(r'^(\s*)([-*+] |\d+[.)] )')


def flush(block, out, width):
    if not block:
        return
    first = block[0]
    m = ITEM.match(first)
    lead, marker = (m.group(1), m.group(2)) if m else (re.match(r'^\s*', first).group(0), '')
    body = ' '.join([first[len(lead) + len(marker):].strip()] + [l.strip() for l in block[1:]])
    out.extend(textwrap.wrap(body, width=width, initial_indent=lead + marker, subsequent_indent=lead + ' ' * len(marker),
                             break_long_words=False, break_on_hyphens=False) or [lead + marker])
    block.clear()


def wrap(text, width=128):
    out, block, fence = [], [], False
    for line in text.split('\n'):
        fixed = line.lstrip().startswith('|') or line.startswith('#') or not line.strip()
        if line.lstrip().startswith('```'):
            flush(block, out, width)
            fence = not fence
            out.append(line)
        elif fence or fixed:
            flush(block, out, width)
            out.append(line)
        elif ITEM.match(line):
            flush(block, out, width)
            block.append(line)
        else:
            block.append(line)
    flush(block, out, width)
    return '\n'.join(out)


if __name__ == '__main__':
    p = sys.argv[1]
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    text = open(p).read()
    open(p, 'w').write(wrap(text, w))
